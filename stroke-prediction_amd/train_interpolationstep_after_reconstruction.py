#!/usr/bin/env python3
"""Step learning after phase 1 on the MI355X path (reference ``train_interpolationstep_after_reconstruction.py:8-73``): the
shape CAE from the positional ``caepath`` is frozen, its encoder stack moves into an ``Enc3DStep`` whose 1x1x1 step layers on the clinical
globals are the only trainable parameters, and ``CaeStepLearner`` trains them (Adam lr 1e-3, betas (0.9, 0.999), wd 1e-5).

    python stroke-prediction_amd/train_interpolationstep_after_reconstruction.py /tmp/x_cae1.model --epochs 2 --batchsize 4
"""
import datetime
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stroke_prediction_amd  # noqa: E402,F401
from common import data, metrics, util  # noqa: E402
from common.model.Cae3D import Cae3D, Enc3DStep  # noqa: E402
from learner.CaeStepLearner import CaeStepLearner  # noqa: E402

LABEL_VOLUMES = ['_CBVmap_subset_reg1_downsampled', '_TTDmap_subset_reg1_downsampled',
                 '_FUCT_MAP_T_Samplespace_subset_reg1_downsampled']
IMAGE_VOLUMES = ['_CBV_reg1_downsampled', '_TTD_reg1_downsampled']              # visualisation only in the reference


def build_model(args):
    cae = torch.load(args.caepath, weights_only=False)
    cae.freeze(True)
    side = int(args.xyoriginal * args.xyresample)
    enc = Enc3DStep(size_input_xy=side, size_input_z=args.zsize, channels=args.channelscae, n_ch_global=args.globals, alpha=1.0,
                    dtype=getattr(cae.enc, "compute_dtype", "bf16"))
    enc.encoder = cae.enc.encoder        # the step layers are trained from scratch for the given shape representation
    return Cae3D(enc, cae.dec).cuda()


def train(args):
    cae = build_model(args)
    params = [p for p in cae.parameters() if p.requires_grad]
    print('# optimizing params', sum(p.nelement() for p in params), '/ total: cae', sum(p.nelement() for p in cae.parameters()))
    optimizer = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, args.lrsteps) if args.lrsteps else None
    common = [data.ResamplePlaneXY(args.xyresample)]
    train_tf = common + [data.HemisphericFlip(), data.ElasticDeform(), data.ToTensor()]
    valid_tf = common + [data.ToTensor()]
    ds_train, ds_valid = data.get_stroke_shape_training_data(IMAGE_VOLUMES, LABEL_VOLUMES, train_tf, valid_tf, args.fold,
                                                             args.validsetsize, batchsize=args.batchsize)
    print('Size training set:', len(ds_train.sampler.indices), 'samples | Size validation set:', len(ds_valid.sampler.indices),
          'samples | Capacity batch:', args.batchsize, 'samples')
    learner = CaeStepLearner(ds_train, ds_valid, cae, optimizer, scheduler, n_epochs=args.epochs,
                             path_previous_base=args.inbasepath, path_outputs_base=args.outbasepath,
                             criterion=metrics.BatchDiceLoss([1.0]), verbose=False)
    learner.run_training()
    return learner


if __name__ == '__main__':
    print(datetime.datetime.now())
    train(util.get_args_step_training())
    print(datetime.datetime.now())
