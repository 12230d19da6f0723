"""Data formats on either side of the hot path."""
from .data_utils import (compress_and_save_cloth, crop_rois, decompress_cloth_labels,  # noqa: F401
                         decompress_cloth_segment, flip_rois_, to_onehot_tensor)
