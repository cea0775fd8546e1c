// resize.hip -- the Resize step of the reference's input pipeline on the device, bit for bit.
//
// Replaces (reference, /root/reference): transforms.RandomResizedCrop / transforms.Resize inside
// preprocess.py:71-77 / :21-41 (torchvision -> PIL Image.resize, BILINEAR with PIL's antialiasing support), which costs
// the loader workers 0.64 of their 2.8 ms per image.  PIL resamples 8-bit images in FIXED POINT, two passes (horizontal
// into an 8-bit intermediate, then vertical): for every output index a window [xmin, xmin + n) of the input and n
// coefficients round(k * 2^22) - the coefficients are computed in double on the host, by PIL's own recipe
// (data.resample_table), and travel with the batch - and
//     out = clip8((2^21 + sum_j in[xmin + j] * kk[j]) >> 22)
// is pure integer arithmetic, reproduced here exactly.  The workers ship the uint8 CROP (variable size) instead of the
// resized 224 x 224 image; the horizontal flip (RandomHorizontalFlip follows the resize) is a mirrored output column.
//
// Batch layout: pixels = all crops back to back, HWC uint8; meta[b][8] = {pixel offset (bytes), h, w, flip, offset of the
// horizontal table (int32 units), taps per horizontal entry, offset of the vertical table, taps per vertical entry};
// a table = S entries of {xmin, n, kk[taps]}.  tmp rows: row_off[b] = first row of crop b in the [rows][S][C]
// intermediate.
#include "cn_api_internal.h"

#define RS_BITS 22

__device__ __forceinline__ unsigned char rs_clip8(int v) {
  v >>= RS_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[row_off[b] + y][X][c] for every source row y of crop b; one workgroup per source row
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* pixels, const long long* meta, const int* tables,
                                                      const int* row_owner, const int* row_off, unsigned char* tmp, int S, int C) {
  const int row = blockIdx.x;
  const int b = row_owner[row];
  const long long* m = meta + (size_t)b * 8;
  const int y = row - row_off[b];
  const int w = (int)m[2], taps = (int)m[5];
  const int* tab = tables + m[4];
  const unsigned char* src = pixels + m[0] + (size_t)y * w * C;
  for (int X = threadIdx.x; X < S; X += 256) {
    const int* e = tab + (size_t)X * (2 + taps);
    const int xmin = e[0], n = e[1];
    int acc[4] = {1 << (RS_BITS - 1), 1 << (RS_BITS - 1), 1 << (RS_BITS - 1), 1 << (RS_BITS - 1)};
    for (int j = 0; j < n; ++j) {
      const int k = e[2 + j];
      const unsigned char* p = src + (size_t)(xmin + j) * C;
      for (int c = 0; c < C; ++c) acc[c] += (int)p[c] * k;
    }
    unsigned char* o = tmp + ((size_t)row * S + X) * C;
    for (int c = 0; c < C; ++c) o[c] = rs_clip8(acc[c]);
  }
}

// vertical pass + flip: out[b][Y][X or S-1-X][c]; one workgroup per output row
__global__ __launch_bounds__(256) void resize_v_kernel(const unsigned char* tmp, const long long* meta, const int* tables,
                                                      const int* row_off, unsigned char* out, int S, int C) {
  const int b = blockIdx.x / S, Y = blockIdx.x - b * S;
  const long long* m = meta + (size_t)b * 8;
  const int flip = (int)m[3], taps = (int)m[7];
  const int* e = tables + m[6] + (size_t)Y * (2 + taps);
  const int ymin = e[0], n = e[1];
  const unsigned char* src = tmp + (size_t)(row_off[b] + ymin) * S * C;
  for (int X = threadIdx.x; X < S; X += 256) {
    int acc[4] = {1 << (RS_BITS - 1), 1 << (RS_BITS - 1), 1 << (RS_BITS - 1), 1 << (RS_BITS - 1)};
    for (int j = 0; j < n; ++j) {
      const int k = e[2 + j];
      const unsigned char* p = src + ((size_t)j * S + X) * C;
      for (int c = 0; c < C; ++c) acc[c] += (int)p[c] * k;
    }
    unsigned char* o = out + (((size_t)b * S + Y) * S + (flip ? S - 1 - X : X)) * C;
    for (int c = 0; c < C; ++c) o[c] = rs_clip8(acc[c]);
  }
}

// B crops -> out[B][S][S][C] uint8.  total_rows = sum of the crops' heights (rows of tmp, which holds total_rows * S * C
// bytes); row_owner[total_rows] / row_off[B] as above.  All tables and pixels are device memory.
extern "C" int cn_resize_u8_crops(const unsigned char* pixels, const long long* meta, const int* tables, const int* row_owner,
                                  const int* row_off, unsigned char* tmp, unsigned char* out, int B, int total_rows, int S,
                                  int C, void* stream_) {
  if (pixels == nullptr || meta == nullptr || tables == nullptr || row_owner == nullptr || row_off == nullptr || tmp == nullptr ||
      out == nullptr) { cn_set_error("resize_u8_crops: null operand"); return CN_EINVAL; }
  if (B <= 0 || total_rows <= 0 || S <= 0 || C < 1 || C > 4) { cn_set_error("resize_u8_crops: bad shape"); return CN_ESHAPE; }
  hipStream_t stream = (hipStream_t)stream_;
  CN_LAUNCH(resize_h_kernel, dim3((unsigned)total_rows), dim3(256), stream, pixels, meta, tables, row_owner, row_off, tmp, S, C);
  CN_LAUNCH(resize_v_kernel, dim3((unsigned)(B * S)), dim3(256), stream, (const unsigned char*)tmp, meta, tables, row_off, out, S, C);
  return cn_check_launch("resize_u8_crops");
}
