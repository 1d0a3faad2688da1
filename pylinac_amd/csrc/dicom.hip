// DICOM native (uncompressed) Pixel Data -> typed frames on the device: the step BEFORE the hot path (SURVEY.md section 8 row
// f1; pylinac/core/image.py:1383-1444 `DicomImage.__init__`: `self.metadata.pixel_array` -> optional `.astype(dtype)` ->
// `_rescale_dicom_values` :363-389).  The arithmetic is pydicom's (pinned `pydicom>=2.0,<3` in the reference's
// pyproject.toml:40; its source is absent from /root/reference): pixel_data_handlers/numpy_handler.py `get_pixeldata` =
// `np.frombuffer(PixelData[:expected_len], dtype=pixel_dtype(ds))` with pixel_dtype = '<' or '>' by transfer syntax, 'u' or
// 'i' by PixelRepresentation, BitsAllocated / 8 bytes -- the CONTAINER value, bits above BitsStored included -- reshaped to
// (NumberOfFrames, Rows, Columns); `apply_rescale` (= apply_modality_lut) = `arr.astype(float64) * RescaleSlope`, then
// `+= RescaleIntercept` (two roundings).  `unused_bits` = 1 adds what pydicom >= 3 does by default for native data
// (`correct_unused_bits`): unsigned samples keep their low BitsStored bits, signed samples are sign-extended from bit
// BitsStored - 1.
//
// One launch decodes a BATCH of frames that lie anywhere in one device buffer (whole Part-10 files copied as they are, or a
// multi-frame Pixel Data element): frame f starts at byte d_offsets[f], at ANY alignment.  A lane takes 16 source bytes per
// step as an aligned 4-dword load plus the dword that follows, funnel-shifted by the frame's misalignment (v_alignbit), so
// the stream stays coalesced whatever the file layout; the kernel is a copy: HBM-bound, 2 x the frame bytes (container
// output) or 1 + 8 / BitsAllocated-bytes x (float64 output).
#include "pl_common.h"

namespace {

constexpr int kDcThreads = 256;

struct alignas(4) DcU4 { unsigned x, y, z, w; };

__device__ __forceinline__ unsigned dc_bswap16x2(unsigned v) { return __builtin_amdgcn_perm(v, v, 0x02030001u); }
__device__ __forceinline__ unsigned dc_bswap32(unsigned v) { return __builtin_amdgcn_perm(v, v, 0x00010203u); }

// one container sample (already in little-endian order) -> its value as a signed 64-bit integer
template <int IB>
__device__ __forceinline__ long long dc_value(unsigned raw, bool is_signed, int stored, bool fix_unused) {
  constexpr int BITS = IB * 8;
  unsigned v = raw;
  if (fix_unused && stored < BITS) {
    if (is_signed) {
      const int sh = 32 - stored;
      return (long long)((int)(v << sh) >> sh);
    }
    v &= (1u << stored) - 1u;
    return (long long)v;
  }
  if (is_signed) {
    const int sh = 32 - BITS;
    return (long long)((int)(v << sh) >> sh);
  }
  return (long long)v;
}

template <typename OutT>
__device__ __forceinline__ OutT dc_out(long long v, bool rescale, double slope, double intercept) {
  if constexpr (sizeof(OutT) == 8) {
    double d = (double)v;                                   // arr.astype(np.float64): exact for every container value
    if (rescale) {
      d = d * slope;                                        // two IEEE operations, like numpy's (no FMA: -ffp-contract=off)
      d = d + intercept;
    }
    return d;
  } else {
    return (float)v;                                        // arr.astype(np.float32): RN of the integer, like numpy's cast
  }
}

// IB = bytes per container sample; MODE 0: container output (same width, the bits as stored or with the unused bits fixed),
// 1: float32, 2: float64 (+ optional rescale)
template <int IB, int MODE>
__global__ void __launch_bounds__(kDcThreads)
dicom_decode_kernel(const unsigned char* __restrict__ bytes, int64_t nbytes, const int64_t* __restrict__ offsets,
                    int64_t samples /* per frame */, int is_signed, int stored, int big_endian, int fix_unused, int rescale,
                    double slope, double intercept, void* __restrict__ out, int32_t* __restrict__ status) {
  const int64_t f = blockIdx.y;
  const int64_t off = offsets[f];
  const int64_t frame_bytes = samples * IB;
  // a frame that does not lie inside the buffer is reported, not read (pydicom: "The length of the pixel data in the
  // dataset doesn't match the expected length" -> ValueError)
  const bool inside = off >= 0 && off + frame_bytes <= nbytes;
  if (blockIdx.x == 0 && threadIdx.x == 0) status[f] = inside ? 0 : 1;
  if (!inside) return;
  const unsigned sh = (unsigned)(off & 3) * 8u;
  const unsigned* base = reinterpret_cast<const unsigned*>(bytes + (off & ~(int64_t)3));
  const int64_t last_dword = ((nbytes + 3) >> 2) - 1 - ((off & ~(int64_t)3) >> 2);   // the last dword of the buffer, from base
  const int64_t nvec = frame_bytes >> 4;                   // whole 16-byte steps
  constexpr int SPV = 16 / IB;                             // samples per step
  const bool fixu = fix_unused != 0, sgn = is_signed != 0, resc = rescale != 0;
  auto fix_dword = [&](unsigned d) -> unsigned {           // container output: byte order + unused bits, in place
    if (IB == 2) {
      if (big_endian) d = dc_bswap16x2(d);
      if (fixu && stored < 16) {
        if (sgn) {                                         // sign-extend each half from bit stored - 1
          const int sx = 32 - stored;
          const int lo = (int)(d << (16 + (16 - stored))) >> sx;          // low sample moved to the top, then down
          const int hi = (int)((d >> 16) << sx) >> sx;
          d = ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16);
        } else {
          const unsigned m = (1u << stored) - 1u;
          d &= m | (m << 16);
        }
      }
    } else if (IB == 4) {
      if (big_endian) d = dc_bswap32(d);
      if (fixu && stored < 32) {
        if (sgn) d = (unsigned)(((int)(d << (32 - stored))) >> (32 - stored));
        else d &= (1u << stored) - 1u;
      }
    } else {
      if (fixu && stored < 8) {
        unsigned r = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          unsigned s = (d >> (8 * b)) & 0xffu;
          if (sgn) s = (unsigned)(((int)(s << (32 - stored))) >> (32 - stored)) & 0xffu;
          else s &= (1u << stored) - 1u;
          r |= s << (8 * b);
        }
        d = r;
      }
    }
    return d;
  };
  for (int64_t v = (int64_t)blockIdx.x * kDcThreads + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kDcThreads) {
    const DcU4 q = *reinterpret_cast<const DcU4*>(base + 4 * v);
    unsigned d[4] = {q.x, q.y, q.z, q.w};
    if (sh) {                                              // (wave-uniform: a property of the frame)
      const int64_t nx = 4 * v + 4;
      const unsigned e = base[nx <= last_dword ? nx : last_dword];     // beyond the buffer only bits nothing uses are needed
      d[0] = __builtin_amdgcn_alignbit(d[1], d[0], sh);
      d[1] = __builtin_amdgcn_alignbit(d[2], d[1], sh);
      d[2] = __builtin_amdgcn_alignbit(d[3], d[2], sh);
      d[3] = __builtin_amdgcn_alignbit(e, d[3], sh);
    }
    if constexpr (MODE == 0) {
      DcU4 o{fix_dword(d[0]), fix_dword(d[1]), fix_dword(d[2]), fix_dword(d[3])};
      *reinterpret_cast<DcU4*>(static_cast<unsigned char*>(out) + f * frame_bytes + 16 * v) = o;
    } else {
      using OutT = typename std::conditional<MODE == 1, float, double>::type;
      OutT* o = static_cast<OutT*>(out) + f * samples + v * SPV;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned w = d[k];
        if (IB == 2 && big_endian) w = dc_bswap16x2(w);
        if (IB == 4 && big_endian) w = dc_bswap32(w);
#pragma unroll
        for (int j = 0; j < 4 / IB; ++j) {
          const unsigned raw = IB == 4 ? w : (IB == 2 ? (w >> (16 * j)) & 0xffffu : (w >> (8 * j)) & 0xffu);
          o[k * (4 / IB) + j] = dc_out<OutT>(dc_value<IB>(raw, sgn, stored, fixu), resc, slope, intercept);
        }
      }
    }
  }
  // the frame's last bytes (fewer than 16): one sample per lane of the first workgroup, byte loads
  if (blockIdx.x == 0) {
    const int64_t s0 = nvec * SPV;
    for (int64_t s = s0 + threadIdx.x; s < samples; s += kDcThreads) {
      const unsigned char* p = bytes + off + s * IB;
      unsigned raw = 0;
#pragma unroll
      for (int b = 0; b < IB; ++b) raw |= (unsigned)p[b] << (8 * (big_endian ? IB - 1 - b : b));
      if constexpr (MODE == 0) {
        const long long val = dc_value<IB>(raw, sgn, stored, fixu);
        unsigned char* o = static_cast<unsigned char*>(out) + f * frame_bytes + s * IB;
#pragma unroll
        for (int b = 0; b < IB; ++b) o[b] = (unsigned char)((unsigned long long)val >> (8 * b));
      } else {
        using OutT = typename std::conditional<MODE == 1, float, double>::type;
        static_cast<OutT*>(out)[f * samples + s] = dc_out<OutT>(dc_value<IB>(raw, sgn, stored, fixu), resc, slope, intercept);
      }
    }
  }
}

}  // namespace

extern "C" int pl_dicom_decode(const unsigned char* d_bytes, int64_t nbytes, const int64_t* d_offsets, int64_t n, int rows,
                               int cols, int bits_allocated, int bits_stored, int pixel_representation, int big_endian,
                               int unused_bits, void* d_out, int out_dtype, int rescale, double slope, double intercept,
                               int32_t* d_status, void* stream) {
  PL_REQUIRE(d_bytes && d_offsets && d_out && d_status, "null pointer");
  PL_REQUIRE(((uintptr_t)d_bytes & 3) == 0, "the byte buffer must start on a 4-byte boundary (frames inside it may start anywhere)");
  PL_REQUIRE(n >= 0 && n <= 65535 && rows > 0 && cols > 0 && nbytes >= 0, "bad shape");
  PL_REQUIRE(bits_allocated == 8 || bits_allocated == 16 || bits_allocated == 32, "BitsAllocated 8, 16 or 32");
  PL_REQUIRE(bits_stored >= 1 && bits_stored <= bits_allocated, "1 <= BitsStored <= BitsAllocated");
  PL_REQUIRE(pixel_representation == 0 || pixel_representation == 1, "PixelRepresentation 0 (unsigned) or 1 (two's complement)");
  const int ib = bits_allocated / 8;
  // the container dtype of pydicom's pixel_dtype: 8-bit PL_U8, 16-bit PL_U16 / PL_I16 by PixelRepresentation, 32-bit PL_I32
  // (wider-than-16 unsigned types travel as their same-width signed bits, as everywhere in this ABI; so does int8 as PL_U8)
  const int container = ib == 1 ? PL_U8 : (ib == 2 ? (pixel_representation ? PL_I16 : PL_U16) : PL_I32);
  PL_REQUIRE(out_dtype == container || out_dtype == PL_F32 || out_dtype == PL_F64,
             "output: the container dtype of BitsAllocated / PixelRepresentation, float32 or float64");
  PL_REQUIRE(!rescale || out_dtype == PL_F64, "the rescale is float64 arithmetic");
  if (n == 0) return PL_OK;
  const int64_t samples = (int64_t)rows * cols;
  const int mode = out_dtype == container ? 0 : (out_dtype == PL_F32 ? 1 : 2);
  // the container form stores 16-byte vectors at 4-byte alignment: frames of a byte count that is no multiple of 4 (odd
  // 8- or 16-bit frames) would misalign the next frame's stores
  PL_REQUIRE(mode != 0 || n == 1 || (samples * ib) % 4 == 0, "container output of a batch needs rows * cols * bytes % 4 == 0");
  const int64_t nvec = samples * ib / 16;
  int64_t bx = pl_cdiv(nvec, (int64_t)kDcThreads * 4);
  if (bx < 1) bx = 1;
  if (bx > 4096) bx = 4096;
  const dim3 grid((unsigned)bx, (unsigned)n);
  hipStream_t st = (hipStream_t)stream;
#define DC_LAUNCH(IB, MODE)                                                                                              \
  hipLaunchKernelGGL((dicom_decode_kernel<IB, MODE>), grid, dim3(kDcThreads), 0, st, d_bytes, nbytes, d_offsets, samples, \
                     pixel_representation, bits_stored, big_endian, unused_bits, rescale, slope, intercept, d_out, d_status)
#define DC_MODE(IB)                          \
  if (mode == 0) DC_LAUNCH(IB, 0);           \
  else if (mode == 1) DC_LAUNCH(IB, 1);      \
  else DC_LAUNCH(IB, 2)
  if (ib == 1) { DC_MODE(1); }
  else if (ib == 2) { DC_MODE(2); }
  else { DC_MODE(4); }
#undef DC_MODE
#undef DC_LAUNCH
  return pl_check_launch("pl_dicom_decode");
}
