// Exact-integer matrix-core kernel for the scipy-exact Gaussian on 16-bit frames.
//
// Replaces: scipy.ndimage.gaussian_filter on uint16 / int16 frames as called at
// pylinac/core/array_utils.py:133 (BaseImage.filter(kind="gaussian"), pylinac/core/image.py:695-712).
//
// Contract (gaussian.hip header): per axis out = trunc(S), S = scipy's float64 tap sequence; axis 0 first, into the 16-bit
// plane, then axis 1 on THAT plane.  The register-window kernels (gaussian_rw.hip, one axis per launch) decide trunc(S)
// with a packed-float32 chain on the VALU.  Here the 41-tap sums move to the matrix cores, in EXACT integer arithmetic,
// and both axes run in ONE launch (gauss2d_mm below):
//
//   taps     w_k = wq_k * 2^-40 + e_k,  wq_k = round(w_k * 2^40) written in FIVE balanced base-256 digits d0..d4 (int8)
//   samples  biased value x in [0, 65535] (int16 input: x = v + 32768);  x - 32896 = 256 * hi + lo with
//            hi = x_hi - 128, lo = x_lo - 128: BOTH digits are the raw bytes with the top bit flipped, both int8
//   T        = sum_k wq_k * (x_k - 32896) = sum over digit pairs 256^(a+b) * sum_k d_a[k] * digit_b[x_k]:
//            NINE v_mfma_i32_16x16x64_i8 per 16 x 16 output tile (Toeplitz band of one weight digit x one sample
//            digit plane, K = the 64-sample window that holds the 16 + 2*RAD <= 64 inputs of 16 outputs), int32
//            accumulation: exact, order-free; five accumulators a1..a5 (digit-pair scales 8, 16, .., 40 bits).  The tenth
//            product (low sample digit x lowest weight digit, scale 0) is worth at most 6e-7 of a grey level: it is left
//            out and its bound is part of D
//   S        = 2^-40 * (T + 32896 * sum wq_k) + E,  |E| 2^40 <= D = 65535 * sum |e_k| 2^40 + the dropped product
//            (D = 1.4e6 at sigma = 5, i.e. 1.3e-6 of a grey level)
//
// Decision.  With C' = 32896 * sum wq_k + M (M = 2^23 >= D) folded into the chain's constants, T' = T + C' and a carry
// cascade over the five digit-pair levels, exact by floor(floor(x / a) / b) = floor(x / (a b)):
//   t_k = a_k + (t_{k-1} >> 8) = floor((2^8 a1 + .. + 2^(8k) a_k) / 2^(8k)),     bits 8k .. 8k+7 of T' = t_k & 255
//   floor(S) = t_5 whenever T' mod 2^40 >= 2 M, i.e. whenever bits 24 .. 39 of T' are not all zero: z = (t3 | t4) & 255 != 0
// The additions of the cascade are the MFMAs' own: level k's first MFMA takes (t_{k-1} >> 8) as its C operand (mm_tile).
// Per output the VALU does four arithmetic right shifts, one add and the byte test -- full-rate opcodes only.
// z = 0 (1.5e-5 of the outputs, plus constant / saturated neighbourhoods where S sits 1e-11 from an integer) means
// "undecided": recomputed with scipy's float64 sequence from the plane bytes still in LDS.
//
// Operand layout: v_mfma_i32_16x16x64_i8 pairs byte s of lane (m, g) of A with byte s of lane (n, g) of B (m, n = lane & 15,
// g = lane >> 4) and leaves D[m = 4 * (lane >> 4) + reg][n = lane & 15] (scripts/ubench/mfma_i8.hip checks this on the
// device); because A and B use the SAME slot -> k map, any consistent assignment of window positions to slots is correct:
// slot (g, s) <-> window position k = 16 g + s.
//
// Cost model (scripts/ubench/mfma_valu_settle.hip on the MI355X, profiles/r03a_ubench_mfma_valu_settle.txt): on one SIMD the
// i8 MFMA (16 cycles, SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA) and the VALU (2.3 - 2.4 cycles per full-rate opcode from two
// or more waves) do NOT overlap -- interleaved in one wave, blocked, or on partner waves, their times add.  A tile costs
// 9 MFMA + ~37 VALU (50 with the plane split and addressing), which is what bounds this kernel (not HBM): DESIGN.md section 5.
#include <type_traits>

#include "pl_common.h"

// Timing-attribution switches for scripts/ubench/g2d_variants.hip ONLY (results become wrong): bit 0 drops the MFMAs,
// bit 1 the integer recombination, bit 2 the global loads, bit 3 the per-step barrier, bit 4 the global stores, bit 5 the plane
// split + LDS writes of the input, bit 6 the LDS writes of the axis-0 plane, bit 7 the LDS operand reads, bit 8 the two level-1 MFMAs and every fix-up.
#ifndef PL_G2D_VARIANT
#define PL_G2D_VARIANT 0
#endif

namespace {

constexpr int kMmHalo = 24;                 // window start = first output - 24: 16-byte aligned, covers RAD <= 24
constexpr int kMmMaxRad = 24;

typedef int v4i __attribute__((ext_vector_type(4)));

// Everything a pass needs, computed on the host (mm_make_params) and passed by value.
constexpr int kMmDigits = 5;
constexpr int kMmQ = 40;                      // taps are quantised to 2^-40

struct MmParams {
  // band[d][c][.]: the zero-padded band sequence E_d[p] = digit_d[|p - 39|] (|p - 39| <= RAD, else 0) of weight digit d,
  // shifted left by c bytes, so that a lane whose Toeplitz row starts at p0 = 16 g - i + 15 reads its 16 bytes as four
  // ALIGNED dwords band[d][p0 & 3][(p0 >> 2) .. + 3]
  unsigned band[kMmDigits][4][24];
  int c1;            // (C' mod 2^32) >> 8     -> initial value of level 1   (C' = 32896 * sum wq + M, M = 2^23)
  int c4;            // C' >> 32               -> added where level 4 starts
  int radius;
  double wd[kMmMaxRad + 1];  // float64 taps (offset j) for the exact path, zero beyond radius
};

bool mm_make_params(const double* h_wts /* 2*R+1 taps, centre at R */, int R, MmParams& p) {
  if (R < 1 || R > kMmMaxRad) return false;
  double wmax = 0.0, W = 0.0;
  for (int k = 0; k <= 2 * R; ++k) {
    if (!(h_wts[k] >= 0.0)) return false;          // floor == trunc needs S >= 0 in the biased domain
    wmax = h_wts[k] > wmax ? h_wts[k] : wmax;
    W += h_wts[k];
  }
  if (!(wmax > 0.0) || !(W < 2.0)) return false;
  // Q is FIXED at 40: the integer part of S then IS the top accumulator level and the decision reads whole bytes of the
  // levels below (mm_tile).  Five balanced digits hold |wq| < 127 * 2^32: taps up to 0.49 (sigma >= 0.82).
  constexpr int Q = kMmQ;
  if (__builtin_ldexp(wmax, Q) >= 5.0e11) return false;
  long long wq[2 * kMmMaxRad + 1];
  long long WQ = 0;
  double eq = 0.0;
  signed char dig[kMmDigits][kMmMaxRad + 1];
  for (int k = 0; k <= 2 * R; ++k) {
    wq[k] = (long long)__builtin_llround(__builtin_ldexp(h_wts[k], Q));
    WQ += wq[k];
    eq += __builtin_fabs(h_wts[k] - __builtin_ldexp((double)wq[k], -Q));
  }
  for (int j = 0; j <= R; ++j) {
    if (wq[R - j] != wq[R + j]) return false;      // the band is built from one half: taps must be symmetric
    long long v = wq[R - j];
    for (int d = 0; d < kMmDigits; ++d) {
      const long long lo = ((v + 128) & 255) - 128;
      dig[d][j] = (signed char)lo;
      v = (v - lo) >> 8;
    }
    if (v != 0) return false;
  }
  for (int d = 0; d < kMmDigits; ++d)
    for (int c = 0; c < 4; ++c) {
      unsigned char bytes[96];
      for (int x = 0; x < 96; ++x) {
        const int pidx = x + c - 39;               // E[x + c], centre at 39
        const int dist = pidx < 0 ? -pidx : pidx;
        bytes[x] = dist <= R ? (unsigned char)dig[d][dist] : 0;
      }
      for (int q = 0; q < 24; ++q)
        p.band[d][c][q] = (unsigned)bytes[4 * q] | ((unsigned)bytes[4 * q + 1] << 8) | ((unsigned)bytes[4 * q + 2] << 16) |
                          ((unsigned)bytes[4 * q + 3] << 24);
    }
  // D bounds |2^Q S_real - (T + C)| in units of T: tap rounding 65535 * sum|e_k| 2^Q; scipy's own rounding and the int16
  // bias 32768 (W - 1), ~1e-11 of a grey level; the product of the two LOWEST digit planes, which is not computed
  // (|sum| <= (2R+1) * 128 * 128); the low byte of C, dropped.
  const double D = __builtin_ldexp(65535.0 * eq * (1.0 + 1e-9) + 65536.0 * __builtin_fabs(W - 1.0) + 1e-9, Q) +
                   (double)(2 * R + 1) * 16384.0 + 256.0;
  // M = 2^23 rides on the accumulators' constants: T' = T + C + M.  If T' mod 2^Q >= 2 M, then floor(T' / 2^Q) =
  // floor(S) whatever the error within +-D <= M; "T' mod 2^Q < 2^24" is "bits 24 .. 39 of T' are all zero"
  constexpr long long M = 1LL << 23;
  if (!(D < (double)M)) return false;
  const long long C = 32896LL * WQ + M;                                      // < 2^16 * 2^41
  p.c1 = (int)((C & 0xffffffffLL) >> 8);
  p.c4 = (int)(C >> 32);
  p.radius = R;
  for (int j = 0; j <= kMmMaxRad; ++j) p.wd[j] = j <= R ? h_wts[R - j] : 0.0;
  return true;
}

// the lane's Toeplitz operand of weight digit d: bytes s = 0..15 <-> window position k = 16 g + s, value
// digit_d[|k - i - 24|] for the lane's index i = lane & 15 inside the 16-output tile
__device__ __forceinline__ v4i mm_band_operand(const MmParams& P, int d, int lane) {
  const int p0 = 16 * (lane >> 4) - (lane & 15) + 15;
  const unsigned* src = &P.band[d][p0 & 3][p0 >> 2];
  return v4i{(int)src[0], (int)src[1], (int)src[2], (int)src[3]};
}

// scipy's value for one output from the two byte planes: lo(k), hi(k) = the plane bytes at window offset k - RAD
template <typename F>
__device__ __forceinline__ double mm_exact(F raw /* k in [-R, R] -> actual value as double */, const MmParams& P) {
  const int R = P.radius;
  double a = raw(0) * P.wd[0];
  for (int j = R; j >= 1; --j) a = __builtin_fma(raw(-j) + raw(j), P.wd[j], a);
  const double off = __builtin_fabs(__builtin_amdgcn_fract(__builtin_fabs(a)) - 0.5);
  if (off > 0.5 - 4e-9) {
    a = raw(0) * P.wd[0];
    for (int j = R; j >= 1; --j) a = a + (raw(-j) + raw(j)) * P.wd[j];
  }
  return a;
}

// One tile's result: v = floor(S) per output in the biased domain (0 .. 65535); z = bits 24 .. 39 of T' folded into a
// byte: the output is decided iff z != 0.
struct MmAcc { v4i v, z; };

struct MmConst { v4i c1, c4; };                // level 1's initial value, level 4's added constant: one register quad each
__device__ __forceinline__ MmConst mm_const(const MmParams& P) {
  return MmConst{v4i{P.c1, P.c1, P.c1, P.c1}, v4i{P.c4, P.c4, P.c4, P.c4}};
}

// The nine MFMAs of one tile as a CARRY CHAIN: img_lo / img_hi = the 16 x 64 sample digit planes (as A when IMG_IS_A),
// w[d] = Toeplitz band of weight digit d.  Level k (digit-pair scale 2^(8k)) starts from the level below shifted right by
// eight -- the shifted value IS the MFMA's C operand, so the additions of the cascade
//   t_k = a_k + (t_{k-1} >> 8) = floor((2^8 a1 + .. + 2^(8k) a_k) / 2^(8k)),     bits 8k .. 8k+7 of T' = t_k & 255
// cost nothing: per output the VALU does four arithmetic shifts, one add (the part of the constant that does not fit
// level 1's 32 bits) and the byte test.  floor(floor(x / a) / b) = floor(x / (a b)) keeps every level exact.  The low
// sample digit x lowest weight digit product (scale 0) is below the decision's resolution: left out, its bound is in D.
template <bool IMG_IS_A>
__device__ __forceinline__ MmAcc mm_tile(v4i img_lo, v4i img_hi, const v4i (&w)[kMmDigits], const MmConst& K) {
  auto mm = [&](v4i img, v4i band, v4i c) -> v4i {
    if (PL_G2D_VARIANT & 1) return c;              // no matrix instruction
    return IMG_IS_A ? __builtin_amdgcn_mfma_i32_16x16x64_i8(img, band, c, 0, 0, 0)
                    : __builtin_amdgcn_mfma_i32_16x16x64_i8(band, img, c, 0, 0, 0);
  };
  v4i t = mm(img_lo, w[1], K.c1);                  // level 1
  t = mm(img_hi, w[0], t);
  t = mm(img_lo, w[2], t >> 8);                    // level 2
  t = mm(img_hi, w[1], t);
  v4i t3 = mm(img_lo, w[3], t >> 8);               // level 3
  t3 = mm(img_hi, w[2], t3);
  v4i t4 = mm(img_lo, w[4], (t3 >> 8) + K.c4);     // level 4
  t4 = mm(img_hi, w[3], t4);
  MmAcc r;
  r.v = mm(img_hi, w[4], t4 >> 8);                 // level 5 = floor(T' / 2^40)
  r.z = (t3 | t4) & 255;
  return r;
}

// NT tiles in lock step, level by level: the MFMA of tile i at level k is followed by the other tiles' MFMAs of that level
// before anything depends on it -- a single tile's chain would stall on the matrix pipe's latency after every level.
template <bool IMG_IS_A, int NT>
__device__ __forceinline__ void mm_tiles(const v4i (&lo)[NT], const v4i (&hi)[NT], const v4i (&w)[kMmDigits], const MmConst& K,
                                         MmAcc (&r)[NT]) {
  auto mm = [&](v4i img, v4i band, v4i c) -> v4i {
    if (PL_G2D_VARIANT & 1) return c;              // no matrix instruction
    return IMG_IS_A ? __builtin_amdgcn_mfma_i32_16x16x64_i8(img, band, c, 0, 0, 0)
                    : __builtin_amdgcn_mfma_i32_16x16x64_i8(band, img, c, 0, 0, 0);
  };
  v4i t[NT], t3[NT], t4[NT];
  if (PL_G2D_VARIANT & 256) {                      // stopwatch only: what would a 7-MFMA tile (no level 1) cost?
#pragma unroll
    for (int i = 0; i < NT; ++i) t[i] = K.c1 + lo[i];
  } else {
#pragma unroll
  for (int i = 0; i < NT; ++i) t[i] = mm(lo[i], w[1], K.c1);                  // level 1
#pragma unroll
  for (int i = 0; i < NT; ++i) t[i] = mm(hi[i], w[0], t[i]);
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) t[i] = mm(lo[i], w[2], t[i] >> 8);             // level 2
#pragma unroll
  for (int i = 0; i < NT; ++i) t[i] = mm(hi[i], w[1], t[i]);
#pragma unroll
  for (int i = 0; i < NT; ++i) t3[i] = mm(lo[i], w[3], t[i] >> 8);            // level 3
#pragma unroll
  for (int i = 0; i < NT; ++i) t3[i] = mm(hi[i], w[2], t3[i]);
#pragma unroll
  for (int i = 0; i < NT; ++i) t4[i] = mm(lo[i], w[4], (t3[i] >> 8) + K.c4);  // level 4
#pragma unroll
  for (int i = 0; i < NT; ++i) t4[i] = mm(hi[i], w[3], t4[i]);
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    r[i].v = mm(hi[i], w[4], t4[i] >> 8);                                      // level 5 = floor(T' / 2^40)
    r[i].z = (t3[i] | t4[i]) & 255;
  }
}

// the actual sample value from its two plane bytes
template <bool SIGNED>
__device__ __forceinline__ double mm_value(unsigned char lo, unsigned char hi) {
  const int x = ((int)(signed char)hi + 128) * 256 + ((int)(signed char)lo + 128);   // biased value 0 .. 65535
  return (double)(x - (SIGNED ? 32768 : 0));
}

// ------------------------------------------------------------------ both axes in ONE kernel: the marching strip
// scipy.ndimage.gaussian_filter on a 16-bit frame is axis 0 into the 16-bit output, then axis 1 on THAT (truncated) plane.
// A workgroup owns a strip of 256 output columns (+ 24 halo columns each side) and marches down a segment of rows, 16
// output rows per step:
//   - a ring of row-group slots of the INPUT digit planes lives in LDS ([slot][column cell: 16 rows x 1 column = 16 bytes]):
//     the 64 window rows of the step's axis-0 tiles.  The row group a later step needs is in flight as buffer loads;
//   - the axis-0 tiles (19 per step: 304 window columns) leave their truncated 16-bit results as digit planes in LDS -- the
//     intermediate frame never goes to HBM;
//   - barrier; the axis-1 tiles (16 per step) read that plane and store the output rows.
// HBM traffic: the frame read once (x 304/256 for the column halo, + 48 rows per segment), written once -- half of the
// two-pass form.  The tile code is straight-line: a wave's tiles run their level chains in lock step (mm_tiles), no branch
// in between (border reflection by arithmetic, stores beyond the segment dropped by the buffer's bounds check, tiles beyond
// a partial strip recompute tile 0, ragged-strip stores chosen per pass).  Undecided outputs (1.5e-5 of the pixels, and
// whole tiles of constant input, which take one wave-uniform evaluation) are recomputed after the pass with scipy's
// float64 sequence from the plane bytes and overwrite what the pass stored.
constexpr int kFCols = 256;                       // output columns per strip
constexpr int kFWin = kFCols + 2 * kMmHalo;       // 304 window columns
constexpr int kFPlane = kFWin * 16;               // bytes of one plane of one row group (= one 16-row axis-0 result plane)
constexpr int kFQuadPitch = (kFWin / 4) * 16;     // byte distance between the cells of columns c and c + 1 (same c >> 2)

// byte offset of column c's cell inside a row-group plane: cells ordered [c & 3][c >> 2] -- the four columns a lane
// splits land 76 cells apart (ds_write_b32: 64 lanes -> 64 banks) and the 16 columns of a tile read conflict-free b128s
__device__ __forceinline__ int f_cell(int c) { return kFQuadPitch * (c & 3) + 16 * (c >> 2); }

struct FQuad { uint2 r[4]; };                     // 4 rows x 4 columns of raw 16-bit samples

// a tile operand: 16 bytes of one plane
__device__ __forceinline__ uint4 f_ldsq(const unsigned char* p) {
  if (PL_G2D_VARIANT & 128) {
    const unsigned a = (unsigned)(uintptr_t)p;
    return uint4{a, a * 3u, a * 5u, a * 7u};
  }
  return *reinterpret_cast<const uint4*>(p);
}

// element q (0..3) of a lane's packed four results
__device__ __forceinline__ void mm_set(uint2& res, int q, unsigned v16) {
  unsigned d = q < 2 ? res.x : res.y;
  d = (q & 1) ? (d & 0x0000ffffu) | (v16 << 16) : (d & 0xffff0000u) | v16;
  if (q < 2) res.x = d; else res.y = d;
}

// true when the lane's two 16-byte operands are one repeated byte each AND every lane of the wave holds the same two bytes
__device__ __forceinline__ bool mm_wave_flat(const uint4& lo, const uint4& hi) {
  const unsigned l0 = lo.x, h0 = hi.x;
  const unsigned lf = __builtin_amdgcn_readfirstlane(l0), hf = __builtin_amdgcn_readfirstlane(h0);   // every lane takes part
  const unsigned d = (lo.y ^ l0) | (lo.z ^ l0) | (lo.w ^ l0) | (hi.y ^ h0) | (hi.z ^ h0) | (hi.w ^ h0) |
                     (__builtin_amdgcn_alignbit(l0, l0, 8) ^ l0) | (__builtin_amdgcn_alignbit(h0, h0, 8) ^ h0) | (lf ^ l0) | (hf ^ h0);
  return __ballot(d != 0u) == 0ull;
}

// four outputs of a lane, branch-free: packed results and ONE flag (some output of the four is undecided)
template <bool SIGNED>
__device__ __forceinline__ uint2 mm_finish_flag(const MmAcc& r, bool& bad) {
  if (PL_G2D_VARIANT & 256) {                      // 7-MFMA stopwatch: results are garbage, nothing is undecided
    bad = false;
    return uint2{__builtin_amdgcn_perm((unsigned)r.v[1], (unsigned)r.v[0], 0x05040100u) ^ (unsigned)r.z[0],
                 __builtin_amdgcn_perm((unsigned)r.v[3], (unsigned)r.v[2], 0x05040100u) ^ (unsigned)r.z[3]};
  }
  if (PL_G2D_VARIANT & 2) {
    bad = false;
    return uint2{(unsigned)(r.v[0] ^ r.z[1]), (unsigned)(r.v[2] ^ r.z[3])};
  }
  unsigned v[4], z[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = (unsigned)r.v[q];
    z[q] = (unsigned)r.z[q];
    if (SIGNED) v[q] = (v[q] + (v[q] < 32768u ? 1u : 0u)) ^ 0x8000u;   // C truncation rounds negative S toward zero
  }
  unsigned zm = z[0] < z[1] ? z[0] : z[1];
  zm = zm < z[2] ? zm : z[2];
  zm = zm < z[3] ? zm : z[3];
  bad = zm == 0u;
  return uint2{__builtin_amdgcn_perm(v[1], v[0], 0x05040100u), __builtin_amdgcn_perm(v[3], v[2], 0x05040100u)};
}

// ------------------------------------------------------------------ eight waves, ONE barrier per step
// Round 2's kernel kept three 4-wave workgroups on a CU, a four-slot ring and one result plane: two barriers per step, and
// per-lane branches around its look-ahead loads that made the compiler wait for every load on the spot.  Here a workgroup
// has eight waves (two workgroups per CU: FOUR waves per SIMD), the input ring has FIVE row-group slots and the axis-0 result
// plane is double-buffered, so that one barrier per step is enough:
//     step s:  A0(s): axis-0 tiles read groups s .. s + 3, results -> plane V[s & 1]
//              W: the group loaded during step s - 1 (group s + 4) goes into slot (s + 4) % 5 -- the slot of group s - 1, dead
//                 since barrier s - 1;   loads of group s + 5 are issued and stay in flight for a whole step
//              barrier s
//              A1(s): axis-1 tiles read V[s & 1], store the output rows
//     A0(s + 1) writes the OTHER plane while slower waves still read V[s & 1]; A0(s + 2) reuses V[s & 1] after barrier s + 1,
//     which every wave reaches only after its A1(s).  W(s + 4) precedes barrier s and is read from A0(s + 1) on.
// The axis-0 plane is tile-major ([tile][row][16 columns], 256 bytes per tile and byte plane): both its ds_write_b32 (64 lanes
// -> 64 consecutive dwords) and the axis-1 ds_read_b128 (lane (j, g) -> tile t + g, row j: 1 KiB contiguous per wave) are
// conflict-free (the row-major plane of the 4-wave kernel, pitch 304, reads 2-way conflicted).
// Work split: the 304 x 16 samples of a row group are 76 column quads x 4 row quads = 304 lane tasks: waves 0 .. 3 take
// column quads 16 v + (lane & 15), wave 4 the twelve quads 64 .. 75; the 19 axis-0 tiles go two each to waves 0 .. 4 and three
// each to waves 5 .. 7 (which load nothing); the 16 axis-1 tiles go pairwise (2 v, 2 v + 1: the permlane16_swap partners).
// One segment per strip where the frame allows: 256 x 1024 x 1024 -> 1024 workgroups = exactly two rounds of the 512
// resident ones, 64 steps after a four-group prologue (the 4-wave kernel: 3072 workgroups, four rounds, 22 steps each).
// Phase stopwatch for scripts/ubench/gauss2d_variants.hip ONLY (-DPL_G2D_TIMING): per wave, s_memtime totals of the four
// phases of a step (W + load issue, axis 0, barrier wait, axis 1) -> g2d_dbg[workgroup][wave][4].
#ifndef PL_G2D_TIMING
#define PL_G2D_TIMING 0
#endif
#if PL_G2D_TIMING
__device__ unsigned long long g2d_dbg[4096 * 8 * 4];
#define PL_G2D_STAMP(k) do { const long long t_ = clock64(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define PL_G2D_STAMP(k) do { } while (0)
#endif
constexpr int kGThreads = 512;
constexpr int kGWaves = kGThreads / PL_WAVE;
constexpr int kGSlots = 5;
constexpr int kGInLo = 0;
constexpr int kGInHi = kGInLo + kGSlots * kFPlane;
constexpr int kGVLo = kGInHi + kGSlots * kFPlane;     // two buffers
constexpr int kGVHi = kGVLo + 2 * kFPlane;            // two buffers
constexpr int kGLds = kGVHi + 2 * kFPlane;            // 68096 bytes: two workgroups per CU

template <typename T>
__global__ void __launch_bounds__(kGThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
gauss2d_mm(const T* __restrict__ in, T* __restrict__ out, int h, int w, int strips, int segs, int seg_rows, const MmParams P) {
  constexpr bool kSigned = (T)-1 < (T)0;
  constexpr unsigned kHiFlip = kSigned ? 0u : 0x80808080u;     // int16: the signed high byte already is x_hi - 128
  __shared__ __attribute__((aligned(16))) unsigned char s_mem[kGLds];

  unsigned id = pl_xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % strips;
  id /= strips;
  const int rt = id % segs;
  const size_t frame = id / segs;
  const int tid = threadIdx.x;
  const int lane = tid & (PL_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / PL_WAVE);
  const int j = lane & 15, g = lane >> 4;
  const int c0 = ct * kFCols, r_begin = rt * seg_rows;
  const int r_end = r_begin + seg_rows < h ? r_begin + seg_rows : h;
  const int nsteps = (r_end - r_begin + 15) / 16;
  if (nsteps <= 0) return;
  const T* f = in + frame * (size_t)h * w;
  const int wcols = w - c0 < kFCols ? w - c0 : kFCols;      // output columns of this strip (even; the last strip's may be ragged)
  const int nht = (wcols + 15) / 16;                         // axis-1 tiles (the last one may reach beyond the frame)
  const int nvt = nht + 3;                                   // axis-0 tiles the axis-1 windows reach
  const bool ragged = (wcols & 15) != 0;                     // the last tile's stores are cut at the frame's edge
  const __amdgpu_buffer_rsrc_t src = pl_make_rsrc_bounded(f, (unsigned)h * (unsigned)w * 2u);
  const __amdgpu_buffer_rsrc_t dstb = pl_make_rsrc_bounded(out + (frame * (size_t)h + r_begin) * w, (unsigned)(r_end - r_begin) * (unsigned)w * 2u);

  // ---- loader lanes: one column quad (4 columns) x row quad (4 rows) per lane.  A quad lies wholly inside the frame, wholly
  // outside it (the MIRRORED quad is loaded and its four columns land in reverse) or -- when w % 4 == 2 -- across the right edge
  const int rq = g;
  const int cqx = wave < 4 ? 16 * wave + j : 64 + j;
  const bool loader = (wave < 4 || (wave == 4 && j < 12)) && 4 * cqx < 16 * nvt;
  const int col0 = c0 - kMmHalo + 4 * cqx;
  const bool mir = col0 < 0 || col0 >= w;
  const int mc = col0 < 0 ? -col0 - 4 : (col0 >= w ? 2 * w - 4 - col0 : col0);
  const bool across = col0 < w && col0 + 4 > w;
  const unsigned colb = 2u * (unsigned)mc;
  const unsigned colb_ld = across ? colb - 4u : colb;
  const int qd0 = 16 * cqx + 4 * rq + (mir ? 3 * kFQuadPitch : 0);
  const int qdstep = mir ? -kFQuadPitch : kFQuadPitch;
  const unsigned wb = 2u * (unsigned)w;
  // Loads are UNCONDITIONAL and branch-free (a per-lane branch around a load makes the compiler wait for it on the spot --
  // vmcnt(0) after every load, four serialised round trips per row group instead of one group in flight for a whole step).
  // A quad ACROSS the right edge (w % 4 == 2: columns w - 2, w - 1 and their reflections) loads columns w - 4 .. w - 1 -- eight
  // bytes inside the frame -- and store_quad builds {w - 2, w - 1, w - 1, w - 2} from the upper half.
  auto load_quad = [&](int k, FQuad& x) {
    const int rbase = r_begin - kMmHalo + 16 * k + 4 * rq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r = rbase + i;
      r ^= r >> 31;                                          // -r - 1 below the frame
      const int r2 = 2 * h - 1 - r;
      r = r < r2 ? r : r2;                                   // 2 h - 1 - r above it (one reflection: h >= 64)
      x.r[i] = pl_buffer_load_u64(src, loader ? (unsigned)r * wb + colb_ld : 0x80000000u, 0);   // idle lanes: out of range, no traffic
    }
  };
  const bool any_across = (w & 3) == 2;            // kernel-uniform: only then can a quad lie across the right edge
  auto store_quad = [&](int slot, const FQuad& xin) {
    if (!loader) return;
    FQuad x = xin;
    if (any_across) {                              // columns w - 2, w - 1, then their reflections w - 1, w - 2
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned up = xin.r[i].y;
        x.r[i].x = across ? up : xin.r[i].x;
        x.r[i].y = across ? __builtin_amdgcn_alignbit(up, up, 16) : up;
      }
    }
    unsigned char* base = s_mem + kGInLo + slot * kFPlane + qd0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const unsigned a0 = half ? x.r[0].y : x.r[0].x, a1 = half ? x.r[1].y : x.r[1].x;
      const unsigned a2 = half ? x.r[2].y : x.r[2].x, a3 = half ? x.r[3].y : x.r[3].x;
      const unsigned e0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u);   // even element: {lo r0, lo r1, hi r0, hi r1}
      const unsigned e1 = __builtin_amdgcn_perm(a3, a2, 0x05010400u);
      const unsigned o0 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);   // odd element
      const unsigned o1 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
      unsigned char* pe = base + (2 * half) * qdstep;
      unsigned char* po = base + (2 * half + 1) * qdstep;
      *reinterpret_cast<unsigned*>(pe) = __builtin_amdgcn_perm(e1, e0, 0x05040100u) ^ 0x80808080u;
      *reinterpret_cast<unsigned*>(pe + (kGInHi - kGInLo)) = __builtin_amdgcn_perm(e1, e0, 0x07060302u) ^ kHiFlip;
      *reinterpret_cast<unsigned*>(po) = __builtin_amdgcn_perm(o1, o0, 0x05040100u) ^ 0x80808080u;
      *reinterpret_cast<unsigned*>(po + (kGInHi - kGInLo)) = __builtin_amdgcn_perm(o1, o0, 0x07060302u) ^ kHiFlip;
    }
  };
  // the band operands are vector loads: retire them before anything else is in flight (left pending into the loop, the
  // compiler's wait-count pass would put vmcnt(0) in front of the first MFMAs of every step)
  v4i band[kMmDigits];
#pragma unroll
  for (int d = 0; d < kMmDigits; ++d) band[d] = mm_band_operand(P, d, lane);
  __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0), expcnt / lgkmcnt untouched
  MmConst K = mm_const(P);
  asm volatile("" : "+v"(K.c1), "+v"(K.c4));
  // row groups 0 .. 3 go to LDS now, group 4 stays in flight: step s writes group s + 4 and loads group s + 5
  FQuad nxt;
  {
    FQuad a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) load_quad(k, a[k]);
    load_quad(4, nxt);
#pragma unroll
    for (int k = 0; k < 4; ++k) store_quad(k, a[k]);
  }
  __syncthreads();

  const int lane_cell = f_cell(j);                 // cell of column 16 t + j, less the tile's 64 t bytes
  int slot_g = g;                                  // slot of row group s + g
  int wslot = 4;                                   // slot that receives group s + 4
#if PL_G2D_TIMING
  long long tacc[4] = {0, 0, 0, 0}, tlast = clock64();
#endif
#ifdef PL_G2D_EXP
  // experiments (stopwatch builds only): de-phase the workgroups that share a CU
  if (PL_G2D_EXP & 4) { if ((blockIdx.x >> 8) & 1) { __builtin_amdgcn_s_sleep(50); } }
  if (PL_G2D_EXP & 8) { if (blockIdx.x & 1) { __builtin_amdgcn_s_sleep(50); } }
  if (PL_G2D_EXP & 16) { if ((blockIdx.x >> 3) & 1) { __builtin_amdgcn_s_sleep(50); } }
  if (PL_G2D_EXP & 32) { if ((blockIdx.x >> 9) & 1) { __builtin_amdgcn_s_sleep(50); } }
#endif
  auto step = [&](int s) {
    const int vb = (s & 1) * kFPlane;              // this step's axis-0 result plane
    const int lrow = 16 * s + j;                   // the lane's output row inside the segment, both passes
    const bool row_ok = r_begin + lrow < r_end;

    // ---- axis 0: image (M = column 16 t + m) x Toeplitz (N = output row): lane (j, g) gets columns 16 t + 4 g .. + 3 of row j
    {
      const unsigned char* ain = s_mem + kGInLo + slot_g * kFPlane + lane_cell;
      unsigned char* vout = s_mem + kGVLo + vb + 16 * j + 4 * g;     // tile-major plane: + 256 t
      auto run = [&](auto NT, int t0, int tstep) {
        constexpr int N = decltype(NT)::value;
        auto tile_of = [&](int i) { const int t = t0 + tstep * i; return t < nvt ? t : 0; };
        unsigned badbits = 0;
        v4i lo[N], hi[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const uint4 qlo = f_ldsq(ain + 64 * tile_of(i));
          const uint4 qhi = f_ldsq(ain + 64 * tile_of(i) + (kGInHi - kGInLo));
          lo[i] = v4i{(int)qlo.x, (int)qlo.y, (int)qlo.z, (int)qlo.w};
          hi[i] = v4i{(int)qhi.x, (int)qhi.y, (int)qhi.z, (int)qhi.w};
        }
        MmAcc acc[N];
        mm_tiles<true, N>(lo, hi, band, K, acc);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          bool bad;
          const uint2 res = mm_finish_flag<kSigned>(acc[i], bad);
          badbits |= (bad && row_ok) ? (1u << i) : 0u;
          unsigned char* vd = vout + 256 * tile_of(i);
          *reinterpret_cast<unsigned*>(vd) = __builtin_amdgcn_perm(res.y, res.x, 0x06040200u) ^ 0x80808080u;
          *reinterpret_cast<unsigned*>(vd + (kGVHi - kGVLo)) = __builtin_amdgcn_perm(res.y, res.x, 0x07050301u) ^ kHiFlip;
        }
        if (__ballot(badbits != 0u) == 0ull) return;
        // ---- undecided outputs of the wave's tiles (rare): scipy's float64 sequence from the input plane bytes
        auto sample = [&](int x, int p) {          // window column x, plane row p (0 .. 63) of this step
          const int a = ((s + (p >> 4)) % kGSlots) * kFPlane + f_cell(x) + (p & 15);
          return mm_value<kSigned>(s_mem[kGInLo + a], s_mem[kGInHi + a]);
        };
#pragma unroll 1
        for (int i = 0; i < N; ++i) {
          if (__ballot((badbits >> i) & 1u) == 0ull) continue;
          const int t = tile_of(i);
          unsigned char* vd = vout + 256 * t;
          const uint4 flo = *reinterpret_cast<const uint4*>(ain + 64 * t);
          const uint4 fhi = *reinterpret_cast<const uint4*>(ain + 64 * t + (kGInHi - kGInLo));
          if (mm_wave_flat(flo, fhi)) {            // constant input under the whole tile: one value for every output
            const double c = sample(16 * t, 0);
            const unsigned v = (unsigned short)pl_from_double<T>(mm_exact([&](int) { return c; }, P));
            *reinterpret_cast<unsigned*>(vd) = (0x01010101u * (v & 255u)) ^ 0x80808080u;
            *reinterpret_cast<unsigned*>(vd + (kGVHi - kGVLo)) = (0x01010101u * (v >> 8)) ^ kHiFlip;
          } else {                                 // the tile once more (every lane): WHICH outputs are undecided
            const MmAcc ra = mm_tile<true>(v4i{(int)flo.x, (int)flo.y, (int)flo.z, (int)flo.w}, v4i{(int)fhi.x, (int)fhi.y, (int)fhi.z, (int)fhi.w}, band, K);
            if ((badbits >> i) & 1u) {
#pragma unroll 1
              for (int q = 0; q < 4; ++q) {
                if (ra.z[q] != 0) continue;
                const int x = 16 * t + 4 * g + q;
                const unsigned v = (unsigned short)pl_from_double<T>(mm_exact([&](int k) { return sample(x, kMmHalo + j + k); }, P));
                vd[q] = (unsigned char)((v & 255u) ^ 0x80u);
                vd[q + (kGVHi - kGVLo)] = (unsigned char)((v >> 8) ^ (kHiFlip & 0x80u));
              }
            }
          }
        }
      };
      // tiles 0 .. 9: waves 0 .. 4 (w, w + 5); tiles 10 .. 18: waves 5 .. 7 (10 + (w - 5), + 3, + 6)
      if (wave < 5) run(std::integral_constant<int, 2>{}, wave, 5); else run(std::integral_constant<int, 3>{}, 5 + wave, 3);
    }
    PL_G2D_STAMP(1);
    // W after the axis-0 tiles: the group loaded during the previous step (s + 4) has had a whole step to arrive, and the
    // output stores of the previous axis-1 pass, which the in-order vmcnt makes this wait for as well, an axis-0 pass
    store_quad(wslot, nxt);
    load_quad(s + 5, nxt);
    PL_G2D_STAMP(0);
    __syncthreads();
    PL_G2D_STAMP(2);

    // ---- axis 1: Toeplitz (M = output column) x image (N = row j): lane (j, g) gets columns 16 t + 4 g .. + 3 of row j
    {
      const unsigned char* bin = s_mem + kGVLo + vb + 16 * j + 256 * g;      // tile t + g, row j
      const unsigned doff = row_ok ? ((unsigned)lrow * (unsigned)w + (unsigned)(c0 + 4 * g)) * 2u : 0x80000000u;
      const unsigned doff16 = row_ok ? ((unsigned)lrow * (unsigned)w + (unsigned)(c0 + 8 * (g >> 1))) * 2u : 0x80000000u;
      auto store_cut = [&](unsigned off, int x0, unsigned d0, unsigned d1, unsigned d2, unsigned d3) {
        const unsigned d[4] = {d0, d1, d2, d3};
#pragma unroll
        for (int k = 0; k < 4; ++k) pl_buffer_store_u32(d[k], dstb, x0 + 2 * k < wcols ? off + 4u * (unsigned)k : 0x80000000u, 0);
      };
      auto tile_of = [&](int i) { const int t = 2 * wave + i; return t < nht ? t : 0; };
      constexpr int N = 2;
      unsigned badbits = 0;
      auto hpass = [&](auto RAGGED) {
        v4i lo[N], hi[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const uint4 qlo = f_ldsq(bin + 256 * tile_of(i));
          const uint4 qhi = f_ldsq(bin + 256 * tile_of(i) + (kGVHi - kGVLo));
          lo[i] = v4i{(int)qlo.x, (int)qlo.y, (int)qlo.z, (int)qlo.w};
          hi[i] = v4i{(int)qhi.x, (int)qhi.y, (int)qhi.z, (int)qhi.w};
        }
        MmAcc acc[N];
        mm_tiles<false, N>(lo, hi, band, K, acc);
        bool bad0, bad1;
        const uint2 r0 = mm_finish_flag<kSigned>(acc[0], bad0), r1 = mm_finish_flag<kSigned>(acc[1], bad1);
        badbits |= ((bad0 && row_ok) ? 1u : 0u) | ((bad1 && row_ok) ? 2u : 0u);
        const auto sx = __builtin_amdgcn_permlane16_swap(r0.x, r1.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(r0.y, r1.y, false, false);
        const unsigned tsel = (g & 1) ? (unsigned)tile_of(1) : (unsigned)tile_of(0);
        const uint4 piece{(unsigned)sx[0], (unsigned)sy[0], (unsigned)sx[1], (unsigned)sy[1]};
        if (!decltype(RAGGED)::value) {
          pl_buffer_store_u128(piece, dstb, doff16 + 32u * tsel, 0);
        } else {
          store_cut(doff16 + 32u * tsel, 16 * (int)tsel + 8 * (g >> 1), piece.x, piece.y, piece.z, piece.w);
        }
      };
      if (ragged) hpass(std::true_type{}); else hpass(std::false_type{});
      if (__ballot(badbits != 0u) != 0ull) {
        // the fix-up stores overwrite addresses the pass has just stored from OTHER lanes of this wave: retire those first
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0) (also retires the look-ahead loads: rare path)
        auto store_cut2 = [&](unsigned off, int x0, unsigned d0, unsigned d1) {   // 4 columns from x0, cut at the edge
          pl_buffer_store_u32(d0, dstb, x0 < wcols ? off : 0x80000000u, 0);
          pl_buffer_store_u32(d1, dstb, x0 + 2 < wcols ? off + 4u : 0x80000000u, 0);
        };
        auto sample = [&](int row, int x) {        // row of the step, window column x
          const int a = vb + 256 * (x >> 4) + 16 * row + (x & 15);
          return mm_value<kSigned>(s_mem[kGVLo + a], s_mem[kGVHi + a]);
        };
#pragma unroll 1
        for (int i = 0; i < N; ++i) {
          if (__ballot((badbits >> i) & 1u) == 0ull) continue;
          const int t = tile_of(i);
          const uint4 flo = *reinterpret_cast<const uint4*>(bin + 256 * t);
          const uint4 fhi = *reinterpret_cast<const uint4*>(bin + 256 * t + (kGVHi - kGVLo));
          if (mm_wave_flat(flo, fhi)) {
            const double c = sample(0, 16 * t);
            const unsigned v = (unsigned short)pl_from_double<T>(mm_exact([&](int) { return c; }, P));
            store_cut2(doff + 32u * (unsigned)t, 16 * t + 4 * g, v | (v << 16), v | (v << 16));
          } else {
            const MmAcc ra = mm_tile<false>(v4i{(int)flo.x, (int)flo.y, (int)flo.z, (int)flo.w}, v4i{(int)fhi.x, (int)fhi.y, (int)fhi.z, (int)fhi.w}, band, K);
            if ((badbits >> i) & 1u) {
              bool dummy;
              uint2 res = mm_finish_flag<kSigned>(ra, dummy);
#pragma unroll 1
              for (int q = 0; q < 4; ++q) {
                if (ra.z[q] != 0) continue;
                const int x = kMmHalo + 16 * t + 4 * g + q;
                mm_set(res, q, (unsigned short)pl_from_double<T>(mm_exact([&](int k) { return sample(j, x + k); }, P)));
              }
              store_cut2(doff + 32u * (unsigned)t, 16 * t + 4 * g, res.x, res.y);
            }
          }
        }
      }
    }
    slot_g = slot_g == kGSlots - 1 ? 0 : slot_g + 1;
    wslot = wslot == kGSlots - 1 ? 0 : wslot + 1;
    PL_G2D_STAMP(3);
  };
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) step(s);
#if PL_G2D_TIMING
  if (lane == 0 && blockIdx.x < 4096)
    for (int k = 0; k < 4; ++k) g2d_dbg[(blockIdx.x * 8 + wave) * 4 + k] = (unsigned long long)tacc[k];
#endif
}


// Row segments: a workgroup's cost is its steps plus about three steps' worth of prologue (four row groups, 48 halo rows);
// take the segment count that minimises rounds x (steps + 3) over the resident workgroups, segments of at least 128 rows.
// 256 x 1024 x 1024 with the 8-wave kernel (2 per CU): one segment, 1024 workgroups, two full rounds of 64 + 3 steps.
template <typename T>
int launch_mm2d_t(const T* in, T* out, int64_t n, int h, int w, const MmParams& P, hipStream_t st) {
  const int strips = (int)pl_cdiv(w, kFCols);
  const int64_t resident = (int64_t)pl_cu_count() * 2;
  const int64_t max_segs = h / 128 > 1 ? h / 128 : 1;
  int64_t segs = 1, best = -1;
  for (int64_t cand = 1; cand <= max_segs; ++cand) {
    const int64_t rows = pl_cdiv(pl_cdiv(h, cand), 16) * 16;
    const int64_t wgs = n * strips * pl_cdiv(h, rows);
    const int64_t cost = pl_cdiv(wgs, resident) * (rows / 16 + 3);
    if (best < 0 || cost < best) { best = cost; segs = cand; }
  }
  const int seg_rows = (int)(pl_cdiv(pl_cdiv(h, segs), 16) * 16);
  segs = pl_cdiv(h, seg_rows);
  const int64_t blocks = n * strips * segs;
  if (blocks > 0x7fffffffLL) return -1;
  hipLaunchKernelGGL(gauss2d_mm<T>, dim3((unsigned)blocks), dim3(kGThreads), 0, st, in, out, h, w, strips, (int)segs, seg_rows, P);
  return 0;
}

}  // namespace

// 1 when pl_gauss_mm2d_launch covers this call's shape: both axes in one kernel (16-bit frames, reflect borders)
int pl_gauss_mm2d_covers(const void* in, const void* out, int h, int w, int radius) {
  // h, w >= 64: every row / column the 24-wide halos reach is at most ONE reflection away (the kernel reflects by
  // arithmetic); frames below 2 GiB: 32-bit buffer offsets, 0x80000000 + tile offset stays out of range
  // even widths: dword accesses (rows start on 4-byte boundaries; a strip whose width is not a multiple of 16 takes the
  // dword-by-dword stores)
  if (radius < 1 || radius > kMmMaxRad || h < 64 || w < 64 || (w & 1) || (int64_t)h * w * 2 >= 0x7fff0000LL) return 0;
  return !((reinterpret_cast<uintptr_t>(in) & 3) || (reinterpret_cast<uintptr_t>(out) & 3));
}

// 0 = launched; -1 = shape / alignment / taps not covered (the caller runs the two passes).  wts: HOST memory.
int pl_gauss_mm2d_launch(const void* in, void* out, int is_signed, int64_t n, int h, int w, const double* wts, int radius,
                         hipStream_t st) {
  if (!pl_gauss_mm2d_covers(in, out, h, w, radius)) return -1;
  MmParams P;
  if (!mm_make_params(wts, radius, P)) return -1;
  return is_signed ? launch_mm2d_t<short>((const short*)in, (short*)out, n, h, w, P, st)
                   : launch_mm2d_t<unsigned short>((const unsigned short*)in, (unsigned short*)out, n, h, w, P, st);
}
