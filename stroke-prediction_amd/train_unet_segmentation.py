#!/usr/bin/env python3
"""U-Net segmentation training on the MI355X path.  The reference's ``train_unet_segmentation.py`` cannot run as shipped
(its loader / learner calls do not match their definitions and it reads an undefined ``args.inbasepath``);
this script implements what it sets out to do, in the shape of the self-consistent ``train_shape_reconstruction.py``:
``Unet3D(--channels)`` on random 104 x 104 x 68 patches of the 20-voxel-padded volumes (labels lose the padding again:
the network's valid convolutions shrink 104 -> 64 and 68 -> 28), batch Dice on core + penumbra, Adam(1e-3, betas
(0.99, 0.999), weight decay 1e-5) [+ MultiStepLR], best model written to ``unetpath``.

    python stroke-prediction_amd/train_unet_segmentation.py /tmp/unet.model --epochs 2 --outbasepath /tmp/unet --fusedadam --graph
"""
import datetime
import os
import shutil
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stroke_prediction_amd  # noqa: E402,F401
from common import data, metrics, util  # noqa: E402
from common.model.Unet3D import Unet3D  # noqa: E402
from learner.UnetSegmentationLearner import UnetSegmentationLearner  # noqa: E402

PATCH = (104, 104, 68)                     # train_unet_segmentation.py:42
IMAGE_VOLUMES = ['_CBV_reg1_downsampled', '_TTD_reg1_downsampled']
LABEL_VOLUMES = ['_CBVmap_subset_reg1_downsampled', '_TTDmap_subset_reg1_downsampled']     # core, penumbra


def build_loaders(args):
    pad = args.padding
    chain = lambda: [data.ResamplePlaneXY(args.xyresample), data.HemisphericFlipFixedToCaseId(split_id=args.hemisflipid),
                     data.PadImages(pad[0], pad[1], pad[2], pad_value=0), data.RandomPatch(*PATCH, pad[0], pad[1], pad[2]),
                     data.ToTensor()]
    loaders = data.get_stroke_shape_training_data(IMAGE_VOLUMES, LABEL_VOLUMES, chain(), chain(), args.fold, args.validsetsize,
                                                  seed=args.seed, batchsize=args.batchsize)
    print('Size training set:', len(loaders[0].sampler.indices), 'samples | Size validation set:', len(loaders[1].sampler.indices),
          'samples | Capacity batch:', args.batchsize, 'samples')
    return loaders


def train(args):
    unet = Unet3D(args.channels, dtype=args.dtype).cuda()
    params = [p for p in unet.parameters() if p.requires_grad]
    print('# optimizing params', sum(p.nelement() for p in params), '/ total: unet', sum(p.nelement() for p in unet.parameters()))
    hyper = dict(lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999))
    if args.fusedadam or args.graph:
        from stroke_prediction_amd.optim import FusedAdam
        optimizer = FusedAdam(params, capturable=args.graph, **hyper)
    else:
        optimizer = torch.optim.Adam(params, **hyper)
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, args.lrsteps) if args.lrsteps else None
    ds_train, ds_valid = build_loaders(args)
    learner = UnetSegmentationLearner(ds_train, ds_valid, unet, optimizer, scheduler, args.epochs, metrics.BatchDiceLoss([1.0]),
                                      path_previous_base=args.inbasepath, path_outputs_base=args.outbasepath, graph=args.graph)
    learner.run_training()
    best = learner.path('save', learner.FNB_MODEL)
    if args.unetpath and os.path.exists(best):
        shutil.copyfile(best, args.unetpath)            # where test_unet_segmentation.py / the SDM baseline look for it
    return learner


if __name__ == '__main__':
    print(datetime.datetime.now())
    train(util.get_args_unet_training())
    print(datetime.datetime.now())
