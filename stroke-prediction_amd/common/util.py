"""Command-line configuration of the reference's scripts (``common/util.py:40-145``): the same parser classes, flag names,
types and defaults -- ``--fold --hemisflipid --validsetsize --seed --xyoriginal --xyresample --zsize --padding --lrsteps``
on every experiment, ``--epochs --batchsize --globals --normalize --inbasepath --outbasepath --steplearning`` for the CAE,
positional ``unetpath`` and ``--channels --epochs --outbasepath`` for the U-Net -- so command lines written for the
reference keep working.  Flags that exist only here are marked (MI355X build).  ``get_vis_samples`` (matplotlib sample
grids, util.py:8-34) belongs to the visualisation hooks, which are no-ops in this build.
"""
import argparse

_UNET_CHANNELS = [2, 16, 32, 64, 32, 16, 32, 2]
_CAE_CHANNELS = [1, 16, 24, 32, 100, 200, 1]

# (flag, keyword arguments) per parser family; the reference declares --xyresample as int with default 0.5 (util.py:50):
# a float is what every caller passes and computes with, so it is parsed as one here
_EXPERIMENT = [
    ("--fold", dict(type=int, nargs="+", default=list(range(29)), help="Fold case indices (internal indices, not case numbers on disk)")),
    ("--hemisflipid", dict(type=float, default=15, help="Case id or greater, at which hemispheric flip is applied")),
    ("--validsetsize", dict(type=float, default=0.5, help="Fraction of validation set size")),
    ("--seed", dict(type=int, default=4, help="Seed for any randomization")),
    ("--xyoriginal", dict(type=int, default=256, help="Original size of slices")),
    ("--xyresample", dict(type=float, default=0.5, help="Factor for resampling slices")),
    ("--zsize", dict(type=int, default=28, help="Number of z slices")),
    ("--padding", dict(type=int, nargs="+", default=[20, 20, 20], help="Padding of patches")),
    ("--lrsteps", dict(type=int, nargs="+", default=[], help="MultiStepLR epochs")),
    # MI355X build
    ("--dtype", dict(type=str, default="bf16", choices=["bf16", "f32"], help="(MI355X build) storage / MFMA precision of the HIP path")),
    ("--graph", dict(action="store_true", default=False, help="(MI355X build) replay each training step as one hipGraph")),
    ("--fusedadam", dict(action="store_true", default=False, help="(MI355X build) FusedAdam on the flat buffers instead of torch.optim.Adam")),
]
_CAE = [
    ("--epochs", dict(type=int, default=300, help="Number of epochs")),
    ("--batchsize", dict(type=int, default=4, help="Batch size")),
    ("--globals", dict(type=int, default=5, help="Number of global variables")),
    ("--normalize", dict(type=int, default=10, help="Normalization corresponding to penumbra (hours)")),
    ("--inbasepath", dict(type=str, default=None, help="Path and filename base for loading")),
    ("--outbasepath", dict(type=str, default="/tmp/tmp_out", help="Path and filename base for saving")),
    ("--steplearning", dict(action="store_true", default=False, help="Also learn interpolation step from clinical data")),
]
_UNET = [
    ("unetpath", dict(type=str, help="Path to model of Unet")),
    ("--channels", dict(type=int, nargs="+", default=_UNET_CHANNELS, help="Unet channels")),
    ("--epochs", dict(type=int, default=200, help="Number of epochs")),
    ("--outbasepath", dict(type=str, default="/share/data_zoe1/lucas/Linda_Segmentations/tmp/unet", help="Path and filename base for outputs")),
    # the reference's script reads args.inbasepath and a batch size it hard-codes (train_unet_segmentation.py:12,57): made flags
    ("--inbasepath", dict(type=str, default=None, help="(MI355X build) path and filename base of a training to continue")),
    ("--batchsize", dict(type=int, default=6, help="(MI355X build) batch size (train_unet_segmentation.py:12 hard-codes 6)")),
]
_SDM = [
    ("unet", dict(type=str, help="Path to model of Segmentation Unet")),
    ("--channels", dict(type=int, nargs="+", default=_UNET_CHANNELS, help="Unet channels")),
    ("--downsample", dict(type=int, default=1, help="Downsampling to CAE latent representation size")),
    ("--groundtruth", dict(type=int, default=1, help="Use groundtruth instead of UNet segmentations")),
    ("--visualinspection", dict(type=int, default=0, help="Inspect visually before it is saved")),
    ("--outbasepath", dict(type=str, default="/share/data_zoe1/lucas/Linda_Segmentations/tmp/sdm", help="Path and filename base for outputs")),
]


class ExpParser(argparse.ArgumentParser):
    """util.py:40-58; ``parse_args`` prints the namespace like the reference."""
    EXTRA = ()

    def __init__(self):
        super().__init__()
        for flag, kw in list(_EXPERIMENT) + list(self.EXTRA):
            self.add_argument(flag, **kw)

    def parse_args(self, args=None, namespace=None):
        ns = super().parse_args(args, namespace)
        print(ns)
        return ns


class CAEParser(ExpParser):
    EXTRA = _CAE


class UnetParser(ExpParser):
    EXTRA = _UNET


class SDMParser(ExpParser):
    EXTRA = _SDM


def _with(parser, *more):
    for flag, kw in more:
        parser.add_argument(flag, **kw)
    return parser


_CHANNELSCAE = ("--channelscae", dict(type=int, nargs="+", default=_CAE_CHANNELS, help="CAE channels"))
_CAEPATH = ("caepath", dict(type=str, help="Path to previously trained cae phase1 model"))


def get_args_sdm(argv=None):
    return SDMParser().parse_args(argv)


def get_args_shape_training(argv=None):
    return _with(CAEParser(), _CHANNELSCAE).parse_args(argv)


def get_args_step_training(argv=None):
    return _with(CAEParser(), _CAEPATH, _CHANNELSCAE).parse_args(argv)


def get_args_shape_prediction_training(argv=None):
    return _with(CAEParser(), _CAEPATH,
                 ("--channelsenc", dict(type=int, nargs="+", default=_CAE_CHANNELS, help="CAE channels")),
                 ("--initbycae", dict(action="store_true", default=False, help="Init enc weights by cae's enc"))).parse_args(argv)


def get_args_shape_testing(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--path", action="append", type=str, help="Path to model of Shape CAE")
    p.add_argument("--fold", action="append", type=int, nargs="+", help="Fold case indices")
    p.add_argument("--normalize", type=int, default=10, help="Normalization value corresponding to penumbra (hours)")
    p.add_argument("--outbasepath", type=str, default="/share/data_zoe1/lucas/Linda_Segmentations/tmp/shape", help="Path and filename base for outputs")
    p.add_argument("--xyresample", type=float, default=0.5, help="Factor for resampling slices")
    p.add_argument("--padding", type=int, nargs="+", default=[20, 20, 20], help="Padding of patches")
    return p.parse_args(argv)


def get_args_unet_training(argv=None):
    return UnetParser().parse_args(argv)


def get_vis_samples(train_loader, valid_loader):
    """util.py:8-34 picks six samples for the matplotlib grids of ``visualize_epoch``; those hooks are no-ops here."""
    return [], []
