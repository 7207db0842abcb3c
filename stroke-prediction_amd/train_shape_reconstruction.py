#!/usr/bin/env python3
"""Phase-1 shape training on the MI355X path: what the reference's ``train_shape_reconstruction.py:8-79`` does -- a CAE
(``Enc3D`` or ``Enc3DStep`` + ``Dec3D``) trained by ``CaeReconstructionLearner`` with Adam(lr 1e-3, betas (0.9, 0.999),
weight decay 1e-5) [+ MultiStepLR] -- with the same flags (``common/util.py``).  The reference's own script also runs
unchanged against this package (``PYTHONPATH=stroke-prediction_amd python train_shape_reconstruction.py``): this file is the
variant that turns on what only exists here (``--fusedadam``, ``--graph``, ``--dtype``) and falls back to synthetic cases
when the private data set is absent.

    python stroke-prediction_amd/train_shape_reconstruction.py --epochs 2 --batchsize 4 --fusedadam --graph
"""
import datetime
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stroke_prediction_amd  # noqa: E402,F401
from common import data, metrics, util  # noqa: E402
from common.model.Cae3D import Cae3D, Dec3D, Enc3D, Enc3DStep  # noqa: E402
from learner.CaeReconstructionLearner import CaeReconstructionLearner  # noqa: E402

LABEL_VOLUMES = ['_CBVmap_subset_reg1_downsampled', '_TTDmap_subset_reg1_downsampled',
                 '_FUCT_MAP_T_Samplespace_subset_reg1_downsampled']            # core, penumbra, follow-up lesion
IMAGE_VOLUMES = ['_CBV_reg1_downsampled', '_TTD_reg1_downsampled']              # visualisation only in the reference


def build_model(args):
    side = int(args.xyoriginal * args.xyresample)
    kw = dict(size_input_xy=side, size_input_z=args.zsize, channels=args.channelscae, n_ch_global=args.globals, alpha=1.0,
              dtype=args.dtype)
    enc = (Enc3DStep if args.steplearning else Enc3D)(**kw)
    return Cae3D(enc, Dec3D(**kw)).cuda()


def build_optimizer(args, cae):
    params = [p for p in cae.parameters() if p.requires_grad]
    print('# optimizing params', sum(p.nelement() for p in params), '/ total: cae', sum(p.nelement() for p in cae.parameters()))
    hyper = dict(lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    if args.fusedadam or args.graph:
        from stroke_prediction_amd.optim import FusedAdam
        optimizer = FusedAdam(params, capturable=args.graph, **hyper)
    else:
        optimizer = torch.optim.Adam(params, **hyper)
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, args.lrsteps) if args.lrsteps else None
    return optimizer, scheduler


def build_loaders(args):
    resample = [data.ResamplePlaneXY(args.xyresample)]
    # ElasticDeform warps (n, n, d) label volumes on the device; the validation chain only changes the layout
    train_tf = resample + [data.HemisphericFlip(), data.ElasticDeform(), data.ToTensor()]
    valid_tf = resample + [data.ToTensor()]
    use_validation = not args.steplearning
    loaders = data.get_stroke_shape_training_data(IMAGE_VOLUMES, LABEL_VOLUMES, train_tf, valid_tf, args.fold, args.validsetsize,
                                                  seed=args.seed, batchsize=args.batchsize, split=use_validation)
    n_valid = len(loaders[1].sampler.indices) if loaders[1] is not None else 0
    print('Size training set:', len(loaders[0].sampler.indices), 'samples | Size validation set:', n_valid,
          'samples | Capacity batch:', args.batchsize, 'samples')
    return loaders


def train(args):
    cae = build_model(args)
    optimizer, scheduler = build_optimizer(args, cae)
    ds_train, ds_valid = build_loaders(args)
    learner = CaeReconstructionLearner(ds_train, ds_valid, cae, optimizer, scheduler, n_epochs=args.epochs,
                                       path_previous_base=args.inbasepath, path_outputs_base=args.outbasepath,
                                       criterion=metrics.BatchDiceLoss([1.0]), normalization_hours_penumbra=args.normalize,
                                       graph=args.graph)
    learner.run_training()
    return learner


if __name__ == '__main__':
    print(datetime.datetime.now())
    train(util.get_args_shape_training())
    print(datetime.datetime.now())
