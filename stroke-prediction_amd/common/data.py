"""Batch-dict contract of the reference data layer (``common/data.py:18-27``): only the keys and the
channel dimension constant are part of the hot path; the NIfTI dataset / augmentation pipeline is out
of scope (private data set)."""
KEY_CASE_ID = 'case_id'
KEY_CLINICAL_IDX = 'clinical_idx'
KEY_IMAGES = 'images'
KEY_LABELS = 'labels'
KEY_GLOBAL = 'clinical'

DIM_CHANNEL_TORCH3D_5 = 1     # tensors are B x C x D x H x W
