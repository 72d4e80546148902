// Shared epilogue helpers: per-channel (sum, sum of squares) of a kernel's OUTPUT tile, so the next GroupNorm needs no
// separate statistics pass over HBM.  Deterministic: every wave writes its own LDS slot, every workgroup its own partial row.
#pragma once
#include "common.h"

// v[16] = this lane's values for 16 output channels (index 4g+e of a 32x32 MFMA block: channel 8g + 4*(lane>>5) + e).
// Butterfly over the 32 lanes of each half-wave that halves the live values per step (16 shuffles instead of 80).
// Returns, in every lane, the half-wave total for index perm(lane) = b4*8 + b3*4 + b2*2 + b1 (b_k = bit k of the lane id).
__device__ __forceinline__ float reduce16_over32(const float* v, int lane) {
  float a[8], b[4], c[2], d;
  const bool h4 = lane & 16, h3 = lane & 8, h2 = lane & 4, h1 = lane & 2;
#pragma unroll
  for (int k = 0; k < 8; ++k) { const float mine = h4 ? v[k + 8] : v[k], other = h4 ? v[k] : v[k + 8]; a[k] = mine + __shfl_xor(other, 16); }
#pragma unroll
  for (int k = 0; k < 4; ++k) { const float mine = h3 ? a[k + 4] : a[k], other = h3 ? a[k] : a[k + 4]; b[k] = mine + __shfl_xor(other, 8); }
#pragma unroll
  for (int k = 0; k < 2; ++k) { const float mine = h2 ? b[k + 2] : b[k], other = h2 ? b[k] : b[k + 2]; c[k] = mine + __shfl_xor(other, 4); }
  { const float mine = h1 ? c[1] : c[0], other = h1 ? c[0] : c[1]; d = mine + __shfl_xor(other, 2); }
  return d + __shfl_xor(d, 1);
}

// Stores the half-wave totals of (sum, sumsq) for one 32-channel block into this wave's LDS slot stat[channel_local][2]:
// every (channel, statistic) of the block is written by exactly one lane, so the caller can add the slots of the waves
// that share the channels in a fixed order (no float atomics: results must not depend on timing).
__device__ __forceinline__ void stats_block_to_lds(const float* ssum, const float* ssq, float* stat, int cbase, int lane) {
  const float r1 = reduce16_over32(ssum, lane), r2 = reduce16_over32(ssq, lane);
  if ((lane & 1) == 0) {
    const int idx = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    const int c = cbase + 4 * (lane >> 5) + 8 * (idx >> 2) + (idx & 3);
    stat[2 * c] = r1;
    stat[2 * c + 1] = r2;
  }
}
