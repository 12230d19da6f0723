// swapnet_amd -- shared host-side types.  All device tensors are fp32 NHWC.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

namespace swn {

enum Act : int { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_TANH = 3 };
enum PadMode : int { PAD_ZERO = 0, PAD_REFLECT = 1 };

// NHWC view.  `cs` = distance in floats between consecutive pixels (>= C); a channel
// slice of a wider (concat) buffer is a view with p advanced by the channel offset and
// cs = the buffer's full channel count.  C is always a multiple of 4 (pad channels are
// kept at zero), so every pixel row can be moved with 16-byte accesses.
struct TView {
  float* p = nullptr;
  int N = 0, H = 0, W = 0, C = 0, cs = 0;
  size_t pixels() const { return (size_t)N * H * W; }
  TView slice(int c0, int c) const {
    TView v = *this;
    v.p = p + c0;
    v.C = c;
    return v;
  }
};

// im2col gather geometry of one implicit-GEMM launch.  For logical output pixel (oy,ox)
// and tap (kh,kw) the source coordinate in the (optionally x2 nearest-upsampled) input is
//   ye = oy*stride + kh - pad_t ,  xe = ox*stride + kw - pad_l        (extent H<<ups, W<<ups)
// out-of-range taps read 0 (PAD_ZERO) or the reflected pixel (PAD_REFLECT); the source
// pixel is (ye>>ups, xe>>ups).
struct Gather {
  int KH = 1, KW = 1, stride = 1, pad_t = 0, pad_l = 0;
  int pad_mode = PAD_ZERO;
  int ups = 0;
  int Ho = 0, Wo = 0;  // logical output grid
};

// logical output pixel (oy,ox) is stored at (oy*ymul+yoff, ox*xmul+xoff) of the out view
// (sub-pixel phases of a transposed convolution write interleaved positions).
struct OutMap {
  int ymul = 1, yoff = 0, xmul = 1, xoff = 0;
};

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace swn
