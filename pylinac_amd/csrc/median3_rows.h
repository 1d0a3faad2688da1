// Streaming 3x3 median of a 16-bit frame (scipy.ndimage.median_filter(size=3), mode='reflect':
// pylinac/core/array_utils.py:131), shared by the median kernel itself and by the kernels that CONSUME medians without ever
// writing the median plane (Otsu histogram, threshold + column sums: pipeline fusion of pylinac/core/image.py:695-712 with
// :785-800 and pylinac/picketfence.py:747-750).
//
// A lane owns EIGHT consecutive columns (one 16-byte load per row) and slides down ROWS rows; the column to the left / right
// of its block comes from the neighbouring lane (one cross-lane move each; only the first and last lane of a wave fetch theirs
// from memory, with one masked load).  Each new row contributes SORTED horizontal triples (min3 / med3 / max3), kept in registers for three rows;
// median of nine = med3(max3(lows), med3(mids), min3(highs)).  EVERY lane of the wave must call this (cross-lane moves); a
// lane whose block lies beyond the frame (c0 >= w) computes on a clamped address and its values are meaningless.
// Needs w % 8 == 0, 16-byte aligned rows, h > 1.
#pragma once
// (included after pl_common.h by every user)
#ifndef PL_MEDIAN_ROLL
#define PL_MEDIAN_ROLL 1
#endif

// consume(r, m): m[j] = median of column c0 + j of row r (the value itself: sign-extended for int16 frames).
// AHEAD rows are in flight as raw 16-byte loads before their turn (the consumers that run few waves per CU -- one workgroup
// per frame for the Otsu histogram -- would otherwise pay the full memory latency once per row).
// `r0` (first output row) must be the same in every lane of the wave: it is moved to a scalar register here, so the row
// walk -- reflection at the frame's first / last row and the row pointers -- is scalar arithmetic (rounds 1-3 did a general
// reflect with its integer division, and a 64-bit multiply, per lane and row: ~45 of the 180 vector instructions per row).
template <typename T, int ROWS, int AHEAD = 4, typename F>
__device__ __forceinline__ void pl_median3_rows(const T* __restrict__ f, int h, int w, int c0, int lane, int r0_lane, F&& consume) {
  static_assert(sizeof(T) == 2, "16-bit dtypes");
  const int r0 = __builtin_amdgcn_readfirstlane(r0_lane);
  const bool active = c0 < w;
  const unsigned offc = (unsigned)(active ? c0 : 0) * 2u;
  const bool first = c0 == 0, last = c0 + 8 >= w;
  // the column left of lane 0's block / right of lane 63's comes from memory: ONE masked 2-byte load per row serves both
  const bool edge_l = lane == 0 && !first, edge_r = lane == PL_WAVE - 1 && !last && active;
  const bool edge = edge_l || edge_r;
  const unsigned offe = edge_l ? offc - 2u : offc + 16u;
  struct Raw { uint4 q; int e; };
  auto fetch = [&](int r) {                         // row r of the walk; only rows -1 and h are ever reflected INTO a result,
    const int rm = 2 * h - 1 - r;                   // rows beyond (digested, never consumed) just need a valid address
    int rr = r < rm ? r : rm;
    rr = rr > 0 ? rr : 0;
    const char* row = reinterpret_cast<const char*>(f) + (size_t)rr * (size_t)w * 2u;   // scalar
    Raw x;
    x.q = *reinterpret_cast<const uint4*>(row + offc);
    x.e = edge ? (int)*reinterpret_cast<const T*>(row + offe) : 0;
    return x;
  };
  int lo[3][8], mi[3][8], hi[3][8];
  auto digest = [&](const Raw& x, int slot) {
    const unsigned wd[4] = {x.q.x, x.q.y, x.q.z, x.q.w};
    int v[10];   // columns c0-1 .. c0+8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[1 + 2 * k] = (int)(T)(wd[k] & 0xffffu);
      v[2 + 2 * k] = (int)(T)(wd[k] >> 16);
    }
    // lane 0 / lane 63 have no neighbour in the wave: the DPP move leaves them `old` = the value loaded for them
    const int left = __builtin_amdgcn_update_dpp(x.e, v[8], 0x138, 0xf, 0xf, false);    // wave_shr:1
    const int right = __builtin_amdgcn_update_dpp(x.e, v[1], 0x130, 0xf, 0xf, false);   // wave_shl:1
    v[0] = first ? v[1] : left;     // reflect: column -1 -> column 0
    v[9] = last ? v[8] : right;     // column w -> column w-1
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int a = v[j], b = v[j + 1], c = v[j + 2];
      lo[slot][j] = min(min(a, b), c);
      hi[slot][j] = max(max(a, b), c);
      mi[slot][j] = pl_smed3(a, b, c);
    }
  };
  auto emit = [&](int k) {                          // the output row of walk step k (>= 2)
    const int r = r0 + k - 2;
    if (r >= h) return;                             // wave-uniform
    int m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      m[j] = pl_smed3(max(max(lo[0][j], lo[1][j]), lo[2][j]), pl_smed3(mi[0][j], mi[1][j], mi[2][j]),
                      min(min(hi[0][j], hi[1][j]), hi[2][j]));
    consume(r, m);
  };
  Raw ring[AHEAD];
#pragma unroll
  for (int k = 0; k < AHEAD; ++k) ring[k] = fetch(r0 - 1 + k);
#if PL_MEDIAN_ROLL
  // Rolled walk (the default; -DPL_MEDIAN_ROLL=0 builds the straight-line one): the body below is kChunk rows -- a multiple of 3
  // (the sorted-triple slots) and of AHEAD (the load ring), so every slot index in it is a compile-time constant -- inside a real
  // loop; the ROWS % kChunk rows that remain follow unrolled.  The straight-line form of 34 rows is 40 KB of code per
  // instantiation, and that is what made the two median stages box-dependent: on the evidence boxes where they ran 1.3x slower
  // (0.227 / 0.253 ms against 0.173 / 0.191) the rolled form runs at the fast boxes' rate (0.174 / 0.202: A/B of two library
  // builds on one such box) -- an instruction-cache effect, not the memory system as first assumed.
  constexpr int kChunk = (AHEAD % 3 == 0) ? AHEAD : 3 * AHEAD;
  auto step = [&](int k, int ring_slot, int tri_slot) {      // k may be a run-time value; the two slots are constants
    const Raw cur = ring[ring_slot];
    ring[ring_slot] = fetch(r0 - 1 + k + AHEAD);    // rows past the walk are clamped to a valid address and never consumed
    digest(cur, tri_slot);
  };
  step(0, 0 % AHEAD, 0);
  step(1, 1 % AHEAD, 1);
  constexpr int kMain = (ROWS / kChunk) * kChunk;
#pragma unroll 1
  for (int kb = 2; kb < 2 + kMain; kb += kChunk) {
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      step(kb + j, (2 + j) % AHEAD, (2 + j) % 3);   // kb - 2 is a multiple of kChunk: (kb + j) % AHEAD == (2 + j) % AHEAD, same mod 3
      emit(kb + j);
    }
  }
#pragma unroll
  for (int k = 2 + kMain; k < ROWS + 2; ++k) {
    step(k, k % AHEAD, k % 3);
    emit(k);
  }
#else
#pragma unroll
  for (int k = 0; k < ROWS + 2; ++k) {              // walk row k = frame row r0 - 1 + k
    const Raw cur = ring[k % AHEAD];
    if (k + AHEAD < ROWS + 2) ring[k % AHEAD] = fetch(r0 - 1 + k + AHEAD);
    digest(cur, k % 3);
    if (k < 2) continue;
    emit(k);
  }
#endif
}

// the 16-bit patterns of two values in one dword (low half = a)
__device__ __forceinline__ unsigned pl_pack16(int a, int b) { return ((unsigned)a & 0xffffu) | ((unsigned)b << 16); }

// 1 when pl_median3_rows serves this frame geometry
static inline bool pl_median3_rows_covers(const void* in, int h, int w) {
  return h > 1 && w >= 8 && (w & 7) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (((size_t)h * w) & 7) == 0;
}
