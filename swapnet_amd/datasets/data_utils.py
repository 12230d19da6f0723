"""The cloth-segmentation wire format of the reference (datasets/data_utils.py:298-343): a scipy
CSC `.npz` of integer labels (background 0 is simply not stored), written by warp inference and
read by the texture stage (inference.py:140-149,169-180).

File compatibility is exact in both directions (tests/test_data_formats.py reads a file written by
the reference's `compress_and_save_cloth` and the reference-format loader reads ours).  The dense
19-channel fp32 one-hot never has to cross PCIe: `decompress_cloth_labels` hands the integer map
(1 byte/pixel worth of information) to `WarpModel/TextureModel.set_input`, and the expansion
(`swn_model_set_input_labels` / `swn_op_labels_to_onehot`) and the inverse argmax
(`swn_op_argmax_labels`) run on the device, bit-identically to the host functions below."""
import numpy as np
import torch
from scipy import sparse
from scipy.sparse import load_npz

from ..util import decode_labels


def decompress_cloth_labels(fname):
    """The stored label map as a dense int32 (H, W) array."""
    try:
        data_sparse = load_npz(fname)
    except Exception:
        print("Could not decompress cloth segment:", fname)
        raise
    return np.asarray(data_sparse.todense(), dtype=np.int32)


def to_onehot_tensor(sp_matrix, n_labels, device=None):
    """Sparse 2-D label matrix -> one-hot float tensor (n_labels, H, W); entries equal to 0
    (background) give the all-zero vector (reference :330-343: zeros are not stored in the COO
    form, so channel 0 is never set).  With `device` the expansion runs in the HIP library."""
    dense = np.asarray(sp_matrix.todense(), dtype=np.int64)
    if device is not None:
        lab = torch.from_numpy(dense.astype(np.int32))[None]
        return decode_labels.labels_to_onehot(lab, n_labels)[0]
    out = torch.zeros((n_labels,) + dense.shape, dtype=torch.float32)
    lab = torch.from_numpy(dense)
    fg = lab > 0
    if bool((lab[fg] >= n_labels).any()):
        raise RuntimeError("label out of range for n_labels=%d" % n_labels)   # the reference's sparse ctor raises too
    out.scatter_(0, lab.clamp(min=0)[None], fg[None].float())
    return out


def decompress_cloth_segment(fname, n_labels, device=None):
    """Load a cloth segmentation `.npz` -> one-hot tensor (n_labels, H, W)  (reference :298-308)."""
    try:
        data_sparse = load_npz(fname)
    except Exception:
        print("Could not decompress cloth segment:", fname)
        raise
    return to_onehot_tensor(data_sparse, n_labels, device=device)


def compress_and_save_cloth(cloth_tensor, fname):
    """One-hot (or score) tensor (C, H, W) -> argmax label map -> CSC `.npz` (reference :311-327).
    A tensor that already lives on the GPU is reduced there (`swn_op_argmax_labels`, first maximum
    wins like torch.argmax) and only H*W int32 cross PCIe."""
    assert len(cloth_tensor.shape) == 3, "can only compress 1 tensor at a time. remove the preceeding batch size"
    if cloth_tensor.is_cuda:
        max_only = decode_labels.argmax_labels(cloth_tensor[None])[0]
    else:
        max_only = cloth_tensor.argmax(dim=0)
    as_numpy = max_only.cpu().numpy().astype(np.int64)
    sparse.save_npz(fname, sparse.csc_matrix(as_numpy))


# ---- ROI bookkeeping of TextureDataset (reference datasets/data_utils.py:197-295): integer-exact ----
def crop_rois(rois, crop_bounds):
    """Clip `[x1, y1, x2, y2]` rows to the crop window `((x_min, y_min), (x_max, y_max))` and shift
    them into its coordinates (reference :197-234).  numpy arrays and torch tensors; the input is
    not modified; `crop_bounds=None` returns the input itself."""
    if not isinstance(rois, (np.ndarray, torch.Tensor)):
        raise ValueError(f"input must be numpy ndarray or torch Tensor, received {type(rois)}")
    if crop_bounds is None:
        return rois
    (x_min, y_min), (x_max, y_max) = crop_bounds
    if isinstance(rois, np.ndarray):
        xs = np.clip(rois[:, [0, 2]], x_min, x_max - 1) - x_min
        ys = np.clip(rois[:, [1, 3]], y_min, y_max - 1) - y_min
        return np.stack((xs[:, 0], ys[:, 0], xs[:, 1], ys[:, 1]), 1)
    xs = torch.clamp(rois[:, [0, 2]], x_min, x_max - 1) - x_min
    ys = torch.clamp(rois[:, [1, 3]], y_min, y_max - 1) - y_min
    return torch.stack((xs[:, 0], ys[:, 0], xs[:, 1], ys[:, 1]), 1)


def flip_rois_(rois, axis, center):
    """Mirror ROI rows in place about `center` (reference :261-295): axis 0 flips the y pair,
    axis 1 the x pair; min and max swap so each row stays ordered."""
    if axis == 0:
        min_idx, max_idx = -3, -1
    elif axis == 1:
        min_idx, max_idx = -4, -2
    else:
        raise ValueError(f"dim argument must be 0 or 1, received {axis}")
    lo = 2 * center - rois[:, max_idx].clone()
    hi = 2 * center - rois[:, min_idx].clone()
    rois[:, min_idx], rois[:, max_idx] = lo, hi
    return rois
