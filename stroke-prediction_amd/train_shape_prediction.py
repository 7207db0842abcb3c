#!/usr/bin/env python3
"""Phase-2 shape-prediction training on the MI355X path: what the reference's ``train_shape_prediction.py:8-78`` does -- a
NEW ``Enc3D`` (or the CAE's own encoder, ``--initbycae``) trained by ``CaePredictionLearner`` on the U-Net segmentations against
the frozen shape CAE loaded from the positional ``caepath``, Adam(lr 1e-3, betas (0.9, 0.999), weight decay 1e-5) [+ MultiStepLR] -- with
the same flags (``common/util.py:get_args_shape_prediction_training``).  Falls back to synthetic cases when the private data
set is absent; ``--fusedadam`` / ``--dtype`` are additions.

    python stroke-prediction_amd/train_shape_prediction.py /tmp/x_cae1.model --epochs 2 --batchsize 4
"""
import copy
import datetime
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stroke_prediction_amd  # noqa: E402,F401
from common import data, metrics, util  # noqa: E402
from common.model.Cae3D import Enc3D  # noqa: E402
from learner.CaePredictionLearner import CaePredictionLearner  # noqa: E402

MODALITIES = ['_unet_core', '_unet_penu']
LABEL_VOLUMES = ['_CBVmap_subset_reg1_downsampled', '_TTDmap_subset_reg1_downsampled',
                 '_FUCT_MAP_T_Samplespace_subset_reg1_downsampled']


def build_models(args):
    cae = torch.load(args.caepath, weights_only=False)
    cae.freeze(True)
    if args.initbycae:
        enc = copy.deepcopy(cae.enc)          # (the reference loads the file a second time: an independent copy of the encoder)
        enc.freeze(False)
        enc._flat_parent = None
    else:
        side = int(args.xyoriginal * args.xyresample)
        enc = Enc3D(size_input_xy=side, size_input_z=args.zsize, channels=args.channelsenc, n_ch_global=args.globals, alpha=1.0,
                    dtype=getattr(args, "dtype", "bf16"))
    return cae.cuda(), enc.cuda()


def build_optimizer(args, cae, enc):
    params = [p for p in enc.parameters() if p.requires_grad]
    print('# optimizing params', sum(p.nelement() for p in params), '/ total new enc + old cae',
          sum(p.nelement() for p in enc.parameters()) + sum(p.nelement() for p in cae.parameters()))
    hyper = dict(lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    if getattr(args, "fusedadam", False):
        from stroke_prediction_amd.optim import FusedAdam
        optimizer = FusedAdam(params, **hyper)
    else:
        optimizer = torch.optim.Adam(params, **hyper)
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, args.lrsteps) if args.lrsteps else None
    return optimizer, scheduler


def build_loaders(args):
    common = [data.ResamplePlaneXY(args.xyresample), data.HemisphericFlipFixedToCaseId(split_id=args.hemisflipid)]
    train_tf = common + [data.ElasticDeform(apply_to_images=True), data.ToTensor()]
    valid_tf = common + [data.ToTensor()]
    loaders = data.get_stroke_prediction_training_data(MODALITIES, LABEL_VOLUMES, train_tf, valid_tf, args.fold, args.validsetsize,
                                                       batchsize=args.batchsize)
    print('Size training set:', len(loaders[0].sampler.indices), 'samples | Size validation set:', len(loaders[1].sampler.indices),
          'samples | Capacity batch:', args.batchsize, 'samples')
    return loaders


def train(args):
    cae, enc = build_models(args)
    optimizer, scheduler = build_optimizer(args, cae, enc)
    ds_train, ds_valid = build_loaders(args)
    learner = CaePredictionLearner(ds_train, ds_valid, cae, enc, optimizer, scheduler, n_epochs=args.epochs,
                                   path_previous_base=args.inbasepath, path_outputs_base=args.outbasepath,
                                   criterion=metrics.BatchDiceLoss([1.0]))
    learner.run_training()
    return learner


if __name__ == '__main__':
    print(datetime.datetime.now())
    train(util.get_args_shape_prediction_training())
    print(datetime.datetime.now())
