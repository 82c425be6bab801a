"""yolo-from-scratch_amd: MI355X (gfx950) native hot path of KhaledSharif/yolo-from-scratch.

Drop-in for the reference's `train.py` surface (same names, signatures and state-dict keys); the
compute runs in hand-written HIP kernels behind the C ABI of include/yolohip.h (libyolohip.so).
Import as `yolo_from_scratch_amd` (the directory name carries a hyphen).
"""
from ._lib import LIB_PATH, lib as load_library
from .modules import YOLO, ConvBlock, C3, Bottleneck, SPPF, DEFAULT_ANCHORS
from .functional import ciou_loss, decode_predictions, yolo_loss, yolo_loss_multiscale
from .inference import predict, predict_batch, batched_nms, Detector, InferenceSession, assign_targets_gpu
from .training import train_epoch, eval_epoch, HipAdam, HipTrainer
from .hostside import (nms, compute_iou_corners, compute_box_iou, get_lr_lambda, letterbox_resize, YOLODataset,
                       yolo_collate_fn, compute_optimal_anchors, YOLO_SIZES, stack_targets, synthetic_targets,
                       save_checkpoint, load_checkpoint, raw_collate_fn)
from .pipeline import DevicePrefetcher
from .graph import invalidate_folded_weights

__all__ = [
    "YOLO", "ConvBlock", "C3", "Bottleneck", "SPPF", "ciou_loss", "decode_predictions", "yolo_loss",
    "yolo_loss_multiscale", "predict", "predict_batch", "batched_nms", "Detector", "InferenceSession", "assign_targets_gpu", "train_epoch", "eval_epoch", "HipAdam",
    "HipTrainer", "nms", "compute_iou_corners", "compute_box_iou", "get_lr_lambda", "letterbox_resize",
    "YOLODataset", "yolo_collate_fn", "compute_optimal_anchors", "YOLO_SIZES", "stack_targets", "load_library",
    "LIB_PATH", "DEFAULT_ANCHORS", "synthetic_targets", "save_checkpoint", "load_checkpoint", "raw_collate_fn",
    "DevicePrefetcher", "invalidate_folded_weights",
]
