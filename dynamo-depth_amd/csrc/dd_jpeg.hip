// dd_jpeg.hip -- baseline JPEG decoding of a batch of frames on the device (first stage of the
// input side): the reference decodes every frame with PIL in its DataLoader workers (datasets/base_dataset.py:13-18 `pil_loader`,
// called three times per sample at :140-147, kitti_dataset.py:76-90) -- ~5 ms of host time per triplet and 368 KB of decoded
// pixels per frame over PCIe; here the workers only read the files and the ~25 KB of compressed bytes per frame travel.
//
// Arithmetic for arithmetic what libjpeg(-turbo) executes under PIL's defaults, so the result is PIL's, bit for bit
// (tests/test_jpeg.py; oracle/ref_jpeg.py is the numpy restatement):
//   jpeg_huffman_kernel   sequential Huffman entropy decoding (T.81 F.2.2, jdhuff.c) -- inherently serial per image: ONE LANE per
//                         image decodes, the other 63 lanes of its wave stage the compressed bytes into LDS (coalesced), build
//                         the 9-bit look-ahead tables and write finished 8x8 blocks out (coalesced).  ~2-3 ms per 640x192
//                         frame; a batch is 36 frames on 36 waves, on the prefetch stream, under a 45 ms training step.
//   jpeg_idct_kernel      de-quantisation + the ISLOW integer inverse DCT (jidctint.c), one thread per 8x8 block
//   jpeg_color_kernel     "fancy" triangle-filter chroma up-sampling (jdsample.c) + YCbCr -> RGB (jdcolor.c), one thread per pixel,
//                         writing the (n,H,W,3) uint8 buffer that dd_prepare_frames reads
// Supported: 8-bit baseline (SOF0/SOF1 Huffman), one interleaved scan, 1 or 3 components with luma sampling 1x1 / 2x1 / 1x2 / 2x2
// and 1x1 chroma, restart intervals.  Everything else is refused by the host side (hipops/jpeg.py), which then decodes with PIL.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dynamo_hip.h"

namespace dd {

constexpr int JP_LOOK = 9;                       // look-ahead bits of the fast Huffman table
constexpr int JP_CHUNK = 8192;                   // compressed bytes staged in LDS at a time
constexpr int JP_MIN_AHEAD = 512;                // >= worst-case bytes of one block: (27 + 63 * 26) bits, every byte stuffed

__constant__ unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct JpegGeom {
  int ncomp;
  int h[3], v[3];          // sampling factors
  int mcux, mcuy;          // MCUs per row / column
  int bw[3], bh[3];        // blocks per row / column of each component (padded to whole MCUs)
  long long blk_off[3];    // first block of each component inside an image's coefficient array
  long long blocks;        // blocks per image
  long long plane_off[3];  // byte offset of each component's sample plane inside an image's plane area
  long long plane_bytes;   // per image
};

// ---- entropy decoding --------------------------------------------------------------------------------------------------------
struct HuffLds {
  unsigned short look[4][1 << JP_LOOK];     // (length << 8) | symbol for codes of <= JP_LOOK bits, 0 otherwise
  int maxcode[4][18];                       // jdhuff.c derived tables for the longer codes
  int valoff[4][17];                        // valptr[l] - mincode[l]
  unsigned char vals[4][256];
  unsigned char chunk[JP_CHUNK + 16];
  short block[64];
  unsigned char zigzag[64];                 // (a __constant__ table costs a dependent global load per coefficient)
};

__global__ __launch_bounds__(64) void jpeg_huffman_kernel(const unsigned char* __restrict__ data, long long stride, const DDJpegHeader* __restrict__ hdrs,
                                                          JpegGeom geo, short* __restrict__ coef) {
  __shared__ HuffLds S;
  const int img = blockIdx.x, lane = threadIdx.x;
  const DDJpegHeader& hd = hdrs[img];
  const unsigned char* src = data + (long long)img * stride;
  short* out = coef + (long long)img * geo.blocks * 64;

  // ---- derived tables (jpeg_make_d_derived_tbl), built by the wave ----
  for (int i = lane; i < 4 * (1 << JP_LOOK); i += 64) (&S.look[0][0])[i] = 0;
  for (int i = lane; i < 4 * 256; i += 64) (&S.vals[0][0])[i] = hd.vals[i >> 8][i & 255];
  for (int i = lane; i < 64; i += 64) { S.block[i] = 0; S.zigzag[i] = kZigzag[i]; }
  __syncthreads();
  if (lane < 4) {
    const int t = lane;
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      const int n = hd.bits[t][l - 1];
      S.valoff[t][l] = k - code;
      for (int i = 0; i < n; ++i, ++k, ++code) {
        if (l <= JP_LOOK) {
          const int first = code << (JP_LOOK - l);
          for (int f = 0; f < (1 << (JP_LOOK - l)); ++f) S.look[t][first + f] = (unsigned short)((l << 8) | hd.vals[t][k]);
        }
      }
      S.maxcode[t][l] = n ? code - 1 : -1;
      code <<= 1;
    }
    S.maxcode[t][17] = 0x7fffffff;
  }
  __syncthreads();

  // ---- bit reader state (meaningful in lane 0) ----
  int pos = hd.data_offset;                 // next byte of the file to stage
  const int end = hd.data_end;
  int have = 0, rd = 0;                     // bytes staged in S.chunk, read cursor
  unsigned long long bitbuf = 0;
  int nbits = 0;
  bool hit_marker = false;                  // a marker was met: zeros are fed from here on (jdhuff.c does the same)
  int pred[3] = {0, 0, 0};
  int restarts_left = hd.restart_interval;
  const int restart_interval = hd.restart_interval;

  auto fill = [&]() {                       // lane 0 only: top the bit buffer up to > 32 bits
    // four bytes at once when none of them is 0xFF (neither a stuffed byte nor a marker): one LDS read instead of four
    if (!hit_marker && nbits <= 32 && rd + 4 <= have) {
      const unsigned w4 = (unsigned)S.chunk[rd] | ((unsigned)S.chunk[rd + 1] << 8) | ((unsigned)S.chunk[rd + 2] << 16) | ((unsigned)S.chunk[rd + 3] << 24);
      const unsigned ff = (w4 & (w4 >> 4)) & 0x0f0f0f0fu;                       // 0x0f in a byte <=> that byte is 0xFF
      if (((ff + 0x01010101u) & 0x10101010u) == 0) {
        bitbuf = (bitbuf << 32) | (unsigned long long)__builtin_bswap32(w4);
        nbits += 32;
        rd += 4;
      }
    }
    while (nbits <= 48) {
      int b = 0;
      if (!hit_marker && rd < have) {
        b = S.chunk[rd++];
        if (b == 0xFF) {
          const int nx = rd < have ? S.chunk[rd] : 0xD9;
          ++rd;
          if (nx != 0) { hit_marker = true; rd -= 2; b = 0; }      // leave the marker in place for the restart logic
        }
      }
      bitbuf = (bitbuf << 8) | (unsigned long long)b;
      nbits += 8;
    }
  };
  auto getbits = [&](int n) -> int {
    if (n == 0) return 0;
    if (nbits < n) fill();
    const int v = (int)((bitbuf >> (nbits - n)) & ((1ull << n) - 1));
    nbits -= n;
    return v;
  };
  auto decode = [&](int t) -> int {
    if (nbits < 17) fill();                   // (a corrupt stream walks the length search up to l = 17)
    const int peek = (int)((bitbuf >> (nbits - JP_LOOK)) & ((1 << JP_LOOK) - 1));
    const unsigned short e = S.look[t][peek];
    if (e) { nbits -= e >> 8; return e & 255; }
    int l = JP_LOOK + 1;
    int code = (int)((bitbuf >> (nbits - l)) & ((1 << l) - 1));
    while (code > S.maxcode[t][l]) { ++l; code = (int)((bitbuf >> (nbits - l)) & ((1u << l) - 1)); }
    nbits -= l;
    if (l > 16) return 0;
    return S.vals[t][(code + S.valoff[t][l]) & 255];
  };
  auto extend = [](int v, int s) -> int { return (s && v < (1 << (s - 1))) ? v - ((1 << s) - 1) : v; };

  auto stage = [&]() {                      // whole wave: keep at least one worst-case block of compressed bytes staged (the cursor lives in lane 0)
    const int rd0 = __builtin_amdgcn_readfirstlane(rd);
    if (have - rd0 < JP_MIN_AHEAD && pos < end) {
      const int keep = have - rd0;
      // move the unread tail to the front, then append from the file
      unsigned char tail[8];
      for (int base = 0; base < keep; base += 64 * 8) {
        const int i = base + lane * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) tail[j] = (i + j < keep) ? S.chunk[rd0 + i + j] : 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (i + j < keep) S.chunk[i + j] = tail[j];
        __syncthreads();
      }
      const int take = min(JP_CHUNK - keep, end - pos);
      for (int i = lane; i < take; i += 64) S.chunk[keep + i] = src[pos + i];
      pos += take;
      have = keep + take;
      rd = 0;
      __syncthreads();
    }
  };

  const int total_mcus = geo.mcux * geo.mcuy;
  for (int mcu = 0; mcu < total_mcus; ++mcu) {
    const int my = mcu / geo.mcux, mx = mcu - my * geo.mcux;
    // restart interval: discard the bit buffer, step over the RSTn marker, reset the DC predictions
    if (restart_interval && mcu && restarts_left == 0) {
      stage();                              // the RSTn marker may not be staged yet: top the chunk up before looking for it
      if (lane == 0) {
        bitbuf = 0; nbits = 0;
        while (rd + 1 < have && !(S.chunk[rd] == 0xFF && S.chunk[rd + 1] >= 0xD0 && S.chunk[rd + 1] <= 0xD7)) ++rd;
        rd += 2;
        hit_marker = false;
        pred[0] = pred[1] = pred[2] = 0;
      }
      restarts_left = restart_interval;
    }
    if (restart_interval) --restarts_left;
    for (int ci = 0; ci < geo.ncomp; ++ci) {
      const int td = hd.td[ci], ta = 2 + hd.ta[ci];
      for (int by = 0; by < geo.v[ci]; ++by)
        for (int bx = 0; bx < geo.h[ci]; ++bx) {
          stage();
          if (lane == 0) {
            int s = decode(td);
            pred[ci] += extend(getbits(s), s);
            S.block[0] = (short)pred[ci];
            int k = 1;
            while (k < 64) {
              const int rs = decode(ta);
              const int r = rs >> 4;
              s = rs & 15;
              if (s) {
                k += r;
                S.block[S.zigzag[k & 63]] = (short)extend(getbits(s), s);
                ++k;
              } else if (r == 15) {
                k += 16;
              } else {
                break;
              }
            }
          }
          __syncthreads();
          // the finished block goes out with one coalesced 128-byte store, and the staging block is cleared
          const long long blk = geo.blk_off[ci] + (long long)(my * geo.v[ci] + by) * geo.bw[ci] + (mx * geo.h[ci] + bx);
          out[blk * 64 + lane] = S.block[lane];
          S.block[lane] = 0;
          __syncthreads();
        }
    }
  }
}

// ---- jidctint.c (ISLOW): CONST_BITS 13, PASS1_BITS 2 ------------------------------------------------------------------------
__device__ __forceinline__ void idct_1d(const int (&v)[8], int shift_even, int (&o)[8]) {
  int z2 = v[2], z3 = v[6];
  int z1 = (z2 + z3) * 4433;
  int tmp2 = z1 + z3 * (-15137);
  int tmp3 = z1 + z2 * 6270;
  z2 = v[0]; z3 = v[4];
  int tmp0 = (z2 + z3) << shift_even;
  int tmp1 = (z2 - z3) << shift_even;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = v[7]; tmp1 = v[5]; tmp2 = v[3]; tmp3 = v[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * 9633;
  tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
  z1 *= -7373; z2 *= -20995; z3 = z3 * (-16069) + z5; z4 = z4 * (-3196) + z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3; o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
  o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1; o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const short* __restrict__ coef, const DDJpegHeader* __restrict__ hdrs, JpegGeom geo, int n_images,
                                                        unsigned char* __restrict__ planes) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= geo.blocks * n_images) return;
  const int img = (int)(t / geo.blocks);
  const long long b = t - (long long)img * geo.blocks;
  int ci = 0;
  if (geo.ncomp > 1 && b >= geo.blk_off[1]) ci = (geo.ncomp > 2 && b >= geo.blk_off[2]) ? 2 : 1;
  const long long lb = b - geo.blk_off[ci];
  const int brow = (int)(lb / geo.bw[ci]), bcol = (int)(lb - (long long)brow * geo.bw[ci]);
  const unsigned short* qt = hdrs[img].qt[hdrs[img].tq[ci] & 3];
  // the block's 64 coefficients and its quantisation table with eight 16-byte loads each
  int blk[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint4 cv = reinterpret_cast<const uint4*>(coef + t * 64)[r];
    const uint4 qv = reinterpret_cast<const uint4*>(qt)[r];
    const unsigned cw[4] = {cv.x, cv.y, cv.z, cv.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      blk[r][2 * k] = (int)(short)(cw[k] & 0xffffu) * (int)(qw[k] & 0xffffu);
      blk[r][2 * k + 1] = (int)(short)(cw[k] >> 16) * (int)(qw[k] >> 16);
    }
  }
  int ws[8][8];                                            // [row][col] after pass 1
#pragma unroll
  for (int col = 0; col < 8; ++col) {
    int v[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = blk[r][col];
    idct_1d(v, 13, o);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r][col] = descale(o[r], 13 - 2);
  }
  const int pw = geo.bw[ci] * 8;
  unsigned char* dst = planes + (long long)img * geo.plane_bytes + geo.plane_off[ci] + ((long long)brow * 8) * pw + bcol * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int o[8];
    idct_1d(ws[r], 13, o);
    unsigned int lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int s = descale(o[k], 13 + 2 + 3) + 128;
      s = s < 0 ? 0 : (s > 255 ? 255 : s);
      if (k < 4) lo |= (unsigned)s << (8 * k);
      else hi |= (unsigned)s << (8 * (k - 4));
    }
    *reinterpret_cast<uint2*>(dst + (long long)r * pw) = make_uint2(lo, hi);
  }
}

// ---- jdsample.c fancy up-sampling + jdcolor.c ---------------------------------------------------------------------------------
// chroma sample of the up-sampled plane at output pixel (X, Y); cw x ch = the component's TRUE down-sampled size, pw = padded row pitch
__device__ __forceinline__ int chroma_at(const unsigned char* __restrict__ p, int pw, int cw, int ch, int hs, int vs, int X, int Y) {
  if (hs == 1 && vs == 1) return p[(long long)Y * pw + X];
  if (hs == 2 && vs == 1) {                                // h2v1: (3 c + neighbour + {1, 2}) >> 2, edges copied
    const int x = X >> 1, odd = X & 1;
    const unsigned char* r = p + (long long)Y * pw;
    const int c = r[x];
    if ((x == 0 && !odd) || (x == cw - 1 && odd)) return c;
    return odd ? (3 * c + r[x + 1] + 2) >> 2 : (3 * c + r[x - 1] + 1) >> 2;
  }
  if (hs == 1 && vs == 2) {                                // h1v2
    const int y = Y >> 1, odd = Y & 1;
    const int c = p[(long long)y * pw + X];
    const int o = p[(long long)(odd ? min(y + 1, ch - 1) : max(y - 1, 0)) * pw + X];
    return odd ? (3 * c + o + 2) >> 2 : (3 * c + o + 1) >> 2;
  }
  // h2v2: column sums 3 * this row + the nearer other row, then 3 * this column + the nearer other column (biases 8 / 7)
  const int x = X >> 1, y = Y >> 1, oddx = X & 1, oddy = Y & 1;
  const unsigned char* r0 = p + (long long)y * pw;
  const unsigned char* r1 = p + (long long)(oddy ? min(y + 1, ch - 1) : max(y - 1, 0)) * pw;
  const int cs = 3 * r0[x] + r1[x];
  if ((x == 0 && !oddx)) return (cs * 4 + 8) >> 4;
  if ((x == cw - 1 && oddx)) return (cs * 4 + 7) >> 4;
  const int xn = oddx ? x + 1 : x - 1;
  const int ns = 3 * r0[xn] + r1[xn];
  return (cs * 3 + ns + (oddx ? 7 : 8)) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const unsigned char* __restrict__ planes, JpegGeom geo, int n_images, int H, int W,
                                                         unsigned char* __restrict__ rgb) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)n_images * H * W) return;
  const int img = (int)(t / ((long long)H * W));
  const int p = (int)(t - (long long)img * H * W);
  const int Y = p / W, X = p - Y * W;
  const unsigned char* base = planes + (long long)img * geo.plane_bytes;
  const int y = base[geo.plane_off[0] + (long long)Y * (geo.bw[0] * 8) + X];
  int r = y, g = y, b = y;
  if (geo.ncomp == 3) {
    const int hmax = geo.h[0], vmax = geo.v[0];
    const int hs = hmax / geo.h[1], vs = vmax / geo.v[1];
    const int cw = (W * geo.h[1] + hmax - 1) / hmax, ch = (H * geo.v[1] + vmax - 1) / vmax;
    const int cb = chroma_at(base + geo.plane_off[1], geo.bw[1] * 8, cw, ch, hs, vs, X, Y) - 128;
    const int cr = chroma_at(base + geo.plane_off[2], geo.bw[2] * 8, cw, ch, hs, vs, X, Y) - 128;
    r = y + ((91881 * cr + 32768) >> 16);
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    b = y + ((116130 * cb + 32768) >> 16);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
  }
  unsigned char* o = rgb + t * 3;
  o[0] = (unsigned char)r; o[1] = (unsigned char)g; o[2] = (unsigned char)b;
}

static bool jpeg_geometry(int ncomp, const int* h, const int* v, int H, int W, JpegGeom& g) {
  if (!(ncomp == 1 || ncomp == 3) || H < 1 || W < 1 || H > 16384 || W > 16384) return false;
  g.ncomp = ncomp;
  for (int i = 0; i < 3; ++i) { g.h[i] = i < ncomp ? h[i] : 1; g.v[i] = i < ncomp ? v[i] : 1; }
  if (ncomp == 1) { g.h[0] = g.v[0] = 1; }                 // a single-component scan is never interleaved: one block per MCU
  if (ncomp == 3) {
    if (g.h[1] != 1 || g.v[1] != 1 || g.h[2] != 1 || g.v[2] != 1) return false;
    if (!((g.h[0] == 1 || g.h[0] == 2) && (g.v[0] == 1 || g.v[0] == 2))) return false;
  }
  const int hmax = g.h[0], vmax = g.v[0];
  g.mcux = (W + 8 * hmax - 1) / (8 * hmax);
  g.mcuy = (H + 8 * vmax - 1) / (8 * vmax);
  long long blocks = 0, bytes = 0;
  for (int i = 0; i < 3; ++i) {
    g.bw[i] = i < ncomp ? g.mcux * g.h[i] : 0;
    g.bh[i] = i < ncomp ? g.mcuy * g.v[i] : 0;
    g.blk_off[i] = blocks;
    g.plane_off[i] = bytes;
    blocks += (long long)g.bw[i] * g.bh[i];
    bytes += (long long)g.bw[i] * g.bh[i] * 64;
  }
  g.blocks = blocks;
  g.plane_bytes = (bytes + 15) / 16 * 16;
  return true;
}

}  // namespace dd

extern "C" size_t dd_jpeg_workspace_bytes(int n_images, int H, int W, int ncomp, const int* h, const int* v) {
  dd::JpegGeom g;
  if (n_images < 1 || !dd::jpeg_geometry(ncomp, h, v, H, W, g)) return 0;
  return (size_t)n_images * ((size_t)g.blocks * 64 * sizeof(short) + (size_t)g.plane_bytes) + 256;
}

extern "C" int dd_jpeg_decode(const unsigned char* data, long long stride, const DDJpegHeader* headers, int n_images, int H, int W, int ncomp,
                              const int* h, const int* v, unsigned char* rgb, void* workspace, size_t workspace_bytes, void* stream_) {
  using namespace dd;
  JpegGeom g;
  if (!data || !headers || !rgb || !workspace || n_images < 1 || stride < 1 || !jpeg_geometry(ncomp, h, v, H, W, g)) return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_jpeg_workspace_bytes(n_images, H, W, ncomp, h, v)) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  short* coef = static_cast<short*>(workspace);
  unsigned char* planes = reinterpret_cast<unsigned char*>(coef + (size_t)n_images * g.blocks * 64);
  planes = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(planes) + 15) & ~(uintptr_t)15);
  hipLaunchKernelGGL(jpeg_huffman_kernel, dim3(n_images), dim3(64), 0, stream, data, stride, headers, g, coef);
  const long long nb = g.blocks * n_images;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, coef, headers, g, n_images, planes);
  const long long np = (long long)n_images * H * W;
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, planes, g, n_images, H, W, rgb);
  return (int)hipGetLastError();
}
