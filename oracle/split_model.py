"""numpy model of how the ring GEMM kernels form fp32 products on the 16-bit matrix cores (TEST INFRASTRUCTURE, like the
rest of oracle/: imported by tests/ only, never by the product).

Restates swapnet_amd/csrc/conv_gemm.hip:
  * scale_exp / PC_TOP_A / PC_TOP_B      -- the power-of-two operand scale from the operand's amax
  * split8h + conv_precut_kernel (PL 2)  -- x * 2^k = h + l, h = fp16 (A: truncated, B: nearest), l = fp16(x * 2^k - h) nearest
  * conv_fwd_pc_kernel<..., PL = 2>      -- a b ~ h_a h_b + h_a l_b + l_a h_b in fp32, scales removed as two exact factors
  * split8 (three bf16 planes by truncation) and the six-term product of the weight-gradient kernel
A product of two fp16 (bf16) values is exact in fp32, and the MFMA accumulates in fp32, so a float32 matmul of the plane
matrices models the kernel up to summation order.
"""
import numpy as np

PC_TOP_A, PC_TOP_B = 12, 10


def scale_exp(amax, top):
    """k with amax * 2^k in [2^(top-1), 2^top); 0 for a zero / non-finite operand; clamped to +-100 (conv_gemm.hip scale_exp)."""
    amax = np.float32(amax)
    if not (amax > 0) or not np.isfinite(amax):
        return 0
    e = int((amax.view(np.uint32) >> np.uint32(23)) & np.uint32(255)) - 127
    return int(max(-100, min(100, top - 1 - e)))


def _trunc_fp16(x):
    """fp32 -> fp16 toward zero (v_cvt_pkrtz_f16_f32), subnormals kept."""
    h = x.astype(np.float16)                                   # nearest
    hf = h.astype(np.float32)
    over = np.abs(hf) > np.abs(x)                              # rounded away from zero: step one ulp back
    h = np.where(over, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float16)


def planes_fp16(x, k, truncate_h):
    xs = (x.astype(np.float32) * np.float32(2.0) ** np.float32(k)).astype(np.float32)
    h = _trunc_fp16(xs) if truncate_h else xs.astype(np.float16)
    l = (xs - h.astype(np.float32)).astype(np.float16)          # x' - h is exact in fp32
    return h.astype(np.float32), l.astype(np.float32)


def matmul_two_plane(a, b, ka=None, kb=None):
    """C = A @ B the way conv_fwd_pc_kernel<PL = 2> forms it.  ka / kb override the amax-derived scale exponents."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    ka = scale_exp(np.abs(a).max(), PC_TOP_A) if ka is None else ka
    kb = scale_exp(np.abs(b).max(), PC_TOP_B) if kb is None else kb
    ah, al = planes_fp16(a, ka, True)
    bh, bl = planes_fp16(b, kb, False)
    acc = (al @ bh + ah @ bl) + ah @ bh                         # smallest terms first, fp32 accumulate
    return (acc * np.float32(2.0) ** np.float32(-ka)) * np.float32(2.0) ** np.float32(-kb)


def planes_bf16(x):
    """Three bf16 planes by truncation (split8): hi, mid, lo with x = hi + mid + lo + O(2^-24 x)."""
    x = np.asarray(x, np.float32)
    def top16(v):
        return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    hi = top16(x); r = x - hi
    mid = top16(r); s = r - mid
    lo = top16(s)
    return hi, mid, lo


def matmul_three_plane(a, b):
    ah, am, al = planes_bf16(a); bh, bm, bl = planes_bf16(b)
    return ((al @ bh + ah @ bl) + am @ bm) + ((am @ bh + ah @ bm) + ah @ bh)
