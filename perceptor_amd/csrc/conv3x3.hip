// conv3x3 (stride 1, pad 1) as an LDS-halo implicit GEMM for gfx950 MFMA — the dominant kernel of the UNets.
//
// A 512-thread workgroup (8 waves) owns a TH x 32 pixel tile x BN output channels.  Per 64-channel input chunk
// the (TH+2) x 34 halo patch is staged into LDS ONCE and serves all 9 taps (the generic kernel re-gathers it
// per tap: 9x the L2->LDS traffic and 9x the address arithmetic); per tap only the [BN][64] weight tile moves
// (double-buffered, register-prefetched one tap ahead).  Each wave computes 2 image rows (2 x 32 pixels) x 128
// output channels = 2 x 4 MFMA 32x32x16 blocks (128 accumulator VGPRs), 32 MFMAs per tap between barriers.
//   config A: TH = 8,  waves 4 (rows) x 2 (channels), BN = 256   (Cout % 256 == 0)
//   config B: TH = 16, waves 8 x 1,                  BN = 128   (Cout % 128 == 0)
// LDS: patch rows are 128 B (64 channels); 16-byte chunk index XOR (pixel>>1)&7 -> conflict-free ds_read_b128 for any
// tap shift, since a lane group always covers 16 consecutive-modulo-16 patch pixels.
// Fused: nearest-x2 upsample of the input (patch gather), skip-concat (two sources), optional GroupNorm-apply
// (+FiLM) + activation on the patch as it is written to LDS (zero padding stays zero), bias / per-sample bias /
// activation / residual (optionally through a nearest-x2 upsample) / fp32 or 16-bit NHWC output in the epilogue.
#include "common.h"
#include "epilogue.h"
#include "../../include/perceptor_hip.h"

namespace {

constexpr int PW = 34;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// PRO: 0 none, else 1 + PMI_ACT_* of the fused GroupNorm-apply prologue.  EARLY: prefetch the next patch into registers
// under the last tap's MFMAs (only when the register budget allows and there is no second workgroup to hide the latency).
//
// Every global read of the main loop is a raw buffer load whose offset is set out of range for padding pixels (the
// hardware returns zeros): no divergent branch around a load.  With `if (inside) v = load` the compiler lost count of the
// loads in flight at every branch join and fell back to s_waitcnt vmcnt(0) -- right after issuing the next tap's weight
// loads, i.e. one exposed L2 round trip per tap in front of the MFMAs.
// NJ: 32-channel MFMA blocks per wave (4 = 128 output channels; 1 for convs with <= 32 output channels, e.g. the UNet's last conv).
// NWB: weight buffers in LDS.  2: write the next tap's weights after the third k-step, barrier at the end of the tap.  3 (one
// workgroup per CU, LDS permitting): the buffer being written was last read two taps ago, so the only barrier of a tap sits
// right behind the weight stores and the last k-step plus the next tap's first fragment reads run without a stop.
template <typename T, int WM, int WN, int PRO, bool EARLY, int NJ = 4, int NWB = 2>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv3x3_halo_kernel(const pmi_igemm_args a) {
  constexpr int XB = 2;                                // image rows (32-pixel MFMA blocks) per wave
  constexpr int TH = WM * XB;                          // image rows per tile
  constexpr int NT = WM * WN * 64;                     // threads per workgroup
  constexpr int RPI = NT / 8;                          // tile rows staged per pass (8 threads x 16 B per 128-B row)
  constexpr int BN = WN * NJ * 32;
  constexpr int PP = (TH + 2) * PW;                    // patch pixels
  constexpr int NPI = (PP + RPI - 1) / RPI;            // 16-byte patch chunks per thread
  constexpr int NWI = BN / RPI;                        // 16-byte weight chunks per thread per tap
  constexpr int PATCH_BYTES = NPI * RPI * 128;         // padded to whole staging passes: no bounds test on the LDS writes
  constexpr int WBYTES = BN * 128;
  __shared__ __attribute__((aligned(16))) char smem[PATCH_BYTES + NWB * WBYTES];
  char* const wbuf = smem + PATCH_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;

  const int tiles_x = a.W / 32, tiles_y = a.H / TH, tiles_n = (a.N + BN - 1) / BN;
  const int nimg = a.M / (a.H * a.W);
  int logical = xcd_remap(blockIdx.x, nimg * tiles_y * tiles_x * tiles_n);
  const int tn = logical % tiles_n; logical /= tiles_n;
  const int tx = logical % tiles_x; logical /= tiles_x;
  const int ty = logical % tiles_y;
  const int img = logical / tiles_y;
  const int y0 = ty * TH, x0 = tx * 32, n0 = tn * BN;
#ifdef PMI_STAMPS
#define STAMP(k) do { if (tid == 0 && a.ws) ((long long*)a.ws)[(int64_t)blockIdx.x * 8 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
  STAMP(0);

  const int Cin = a.C0 + a.C1;
  const int Hv = a.H, Wv = a.W;                         // conv runs on the (possibly upsampled) H x W grid
  const int sc = tid & 7;                              // 16-byte chunk (8 channels) this thread stages, fixed
  // per-image views of the two input tensors and the weight rows of this tile (32-bit offsets inside each)
  const int64_t img_px = (int64_t)a.Hin * a.Win;
  const u16* const A0i = (const u16*)a.A0 + (int64_t)img * img_px * a.lda0;
  const u16* const A1i = a.A1 ? (const u16*)a.A1 + (int64_t)img * img_px * a.lda1 : A0i;
  const int64_t bytes0 = ((img_px - 1) * a.lda0 + a.C0) * 2;
  const int64_t bytes1 = a.A1 ? ((img_px - 1) * a.lda1 + a.C1) * 2 : 0;
  const int nrows = min(BN, a.N - n0);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc((const u16*)a.B + (int64_t)n0 * a.ldb, ((int64_t)(nrows - 1) * a.ldb + a.K) * 2);

  // ---- patch staging plan: source pixel (inside the image) per staged chunk, -1 = zero padding ----
  int ppix[NPI];
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int pp = (tid >> 3) + RPI * i;
    int off = -1;
    if (pp < PP) {
      const int py = pp / PW, px = pp - py * PW;
      int sy = y0 - 1 + py, sx = x0 - 1 + px;
      if (sy >= 0 && sy < Hv && sx >= 0 && sx < Wv) {
        if (a.up) { sy >>= 1; sx >>= 1; }
        off = sy * a.Win + sx;
      }
    }
    ppix[i] = off;
  }
  uint32_t wvo[NWI];                                   // byte offset of this thread's weight chunks inside the tile's rows
#pragma unroll
  for (int i = 0; i < NWI; ++i) wvo[i] = (uint32_t)(((tid >> 3) + RPI * i) * a.ldb + sc * 8) * 2u;

  uint4 pr[NPI], wr[NWI];
  float ga[8], gb[8];
  auto load_patch = [&](int chunk) {
    const int cbase = chunk * 64;
    const bool second = cbase >= a.C0;                  // wave-uniform: C0 is a multiple of 64
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(second ? A1i : A0i, second ? bytes1 : bytes0);
    const uint32_t ldb2 = (uint32_t)(second ? a.lda1 : a.lda0) * 2u;
    const uint32_t so = (uint32_t)(cbase - (second ? a.C0 : 0)) * 2u;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const uint32_t vo = ppix[i] >= 0 ? (uint32_t)ppix[i] * ldb2 + (uint32_t)sc * 16u : PMI_BUF_OOB;
      pr[i] = buf_load16(rs, vo, so);
    }
    if (PRO) {
      const float* pa = a.pro_a + (int64_t)img * Cin + cbase + sc * 8;
      const float* pb = a.pro_b + (int64_t)img * Cin + cbase + sc * 8;
      *(float4*)ga = *(const float4*)pa; *(float4*)(ga + 4) = *(const float4*)(pa + 4);
      *(float4*)gb = *(const float4*)pb; *(float4*)(gb + 4) = *(const float4*)(pb + 4);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      uint4 v = pr[i];
      if (PRO) {                                        // GroupNorm-apply + activation; zero padding stays zero
        float f[8];
        unpack8<T>(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = act_apply(f[e] * ga[e] + gb[e], PRO - 1);
        v = pack8<T>(f);
        const uint32_t keep = ppix[i] >= 0 ? 0xffffffffu : 0u;
        v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
      }
      *(uint4*)(smem + swz((tid >> 3) + RPI * i, sc)) = v;
    }
  };
  auto load_w = [&](int chunk, int tap) {
    const uint32_t so = (uint32_t)(tap * Cin + chunk * 64) * 2u;
#pragma unroll
    for (int i = 0; i < NWI; ++i) wr[i] = buf_load16(rsrc_w, wvo[i], so);
  };
  auto store_w = [&](int buf) {
    char* wb = wbuf + buf * WBYTES;
#pragma unroll
    for (int i = 0; i < NWI; ++i) *(uint4*)(wb + swz((tid >> 3) + RPI * i, sc)) = wr[i];
  };

  f32x16 acc[XB][NJ];
#pragma unroll
  for (int i = 0; i < XB; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // MFMAs of one tap; the next tap's weights (already travelling global -> registers) are written to the other LDS buffer
  // after the third of the four 16-channel steps, so their ds_write latency hides under the last step's MFMAs and only the
  // barrier itself remains at the end of the tap (+2-9 % over storing after the last step; after the second step: less).
  auto mma_tap = [&](int tap, int cur, int nxt) {
    const int dy = tap / 3, dx = tap - dy * 3;          // patch row/col offset (tap - 1 + halo 1)
    const char* wb = wbuf + cur * WBYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ch = kk * 2 + lhi;
      uint4 xf[XB], wf[NJ];
#pragma unroll
      for (int i = 0; i < XB; ++i) xf[i] = *(const uint4*)(smem + swz((XB * wm + i + dy) * PW + l31 + dx, ch));
#pragma unroll
      for (int j = 0; j < NJ; ++j) wf[j] = *(const uint4*)(wb + swz(wn * NJ * 32 + j * 32 + l31, ch));
#pragma unroll
      for (int i = 0; i < XB; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma32(wf[j], xf[i], acc[i][j]);
      if (kk == 2) {
        __builtin_amdgcn_sched_barrier(0);
        store_w(nxt);
        if (NWB == 3) __syncthreads();     // publishes the next tap's weights; nobody is reading buffer nxt (last used two taps ago)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  const int nchunks = Cin / 64;
  load_patch(0);
  load_w(0, 0);
  store_patch();
  store_w(0);
  __syncthreads();
  STAMP(1);
  int cur = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    // taps 0..7: the next tap's weights travel global -> registers -> the other LDS buffer under this tap's MFMAs
#pragma unroll 1
    for (int tap = 0; tap < 8; ++tap) {
      load_w(chunk, tap + 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the loads in front of the MFMAs (the scheduler sinks them to their use)
      const int nxt = NWB == 3 ? (cur == 2 ? 0 : cur + 1) : cur ^ 1;
      mma_tap(tap, cur, nxt);
      if (NWB == 2) __syncthreads();
      cur = nxt;
    }
    // tap 8 also brings in the next 64-channel patch.  Without a prologue it is prefetched into registers under the MFMAs;
    // with the fused GroupNorm prologue that would exceed 256 VGPRs, so it is loaded after them (the second workgroup
    // on the CU covers the latency).  After the last chunk the weight load re-reads tile (0, 0): harmless, never used.
    const bool more = chunk + 1 < nchunks;
    load_w(more ? chunk + 1 : 0, 0);
    if (EARLY && more) load_patch(chunk + 1);
    __builtin_amdgcn_sched_barrier(0);
    const int nxt8 = NWB == 3 ? (cur == 2 ? 0 : cur + 1) : cur ^ 1;
    mma_tap(8, cur, nxt8);
    if (more) {
      if (!EARLY) load_patch(chunk + 1);
      __syncthreads();          // every wave is done reading the current patch
      store_patch();
    }
    __syncthreads();
    cur = nxt8;
  }

  STAMP(2);
  if constexpr (NJ < 4) {
    // ---- epilogue, few output channels (N <= 32): 4 channels per lane straight from the accumulator layout; with ldd = 8 the
    // 32 pixels of a wave row form one contiguous 1 KB run.  No statistics / nearest-up residual on this path.
    STAMP(2);
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int y = y0 + XB * wm + i, x = x0 + l31;
      const int64_t o = (((int64_t)img * a.H + y) * a.W + x) * a.ldd;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * NJ * 32 + j * 32 + 4 * lhi + 8 * g;
          if (n >= a.N) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[i][j][4 * g + e] * a.alpha;
            if (a.bias) v[e] += a.bias[n + e];
            if (a.nbias) v[e] += a.nbias[(int64_t)img * (a.ldnb ? a.ldnb : a.N) + n + e];
            if (a.act != PMI_ACT_NONE) v[e] = act_apply(v[e], a.act);
          }
          if (a.R) {
            const int64_t ro = (((int64_t)img * a.H + y) * a.W + x) * a.ldr + n;
            if (a.res_f32) {
              const float4 r = *(const float4*)((const float*)a.R + ro);
              v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
            } else {
              const uint2 r = *(const uint2*)((const u16*)a.R + ro);
              v[0] += T::to_f((u16)(r.x & 0xffff)); v[1] += T::to_f((u16)(r.x >> 16));
              v[2] += T::to_f((u16)(r.y & 0xffff)); v[3] += T::to_f((u16)(r.y >> 16));
            }
          }
          if (a.out_f32) *(float4*)((float*)a.D + o + n) = make_float4(v[0], v[1], v[2], v[3]);
          else *(uint2*)((u16*)a.D + o + n) = pack4<T>(v[0], v[1], v[2], v[3]);
        }
    }
    return;
  } else {
  // ---- epilogue ----
  // Accumulator layout: lane = pixel, 4 consecutive channels per register group -- stored directly, a wave writes 32
  // scattered 16-byte pieces per instruction and the address coalescer needs ~12-20 us per tile for it (in-kernel stamps).
  // Instead each wave transposes its tile through LDS (free after the main loop), 64 channels at a time, and writes whole
  // 128-byte lines: 16 bytes per lane, 8 pixels x 128 B per instruction.  The residual is read the same way.
  constexpr int SROW = 144;                    // staged row: 64 channels x 2 B + 16 B pad (16-byte aligned, rows shift by 36 banks)
  float* const stat = (float*)smem;            // [WM][BN][2] per-channel (sum, sumsq) of each wave row of this tile: plain stores,
                                               // summed in a fixed order below (LDS float atomics would make the result depend on timing)
  float* const bsm = stat + WM * 2 * BN;       // [BN] bias + per-sample bias of the tile's channels
  char* const stg = smem + (WM * 2 + 1) * BN * 4 + wid * (64 * SROW);
  for (int c = tid; c < BN; c += NT) {
    const int n = n0 + c;
    float b = 0.f;
    if (n < a.N) {
      if (a.bias) b = a.bias[n];
      if (a.nbias) b += a.nbias[(int64_t)img * (a.ldnb ? a.ldnb : a.N) + n];
    }
    bsm[c] = b;
  }
  __syncthreads();
  const int r8 = lane & 7, rp = lane >> 3;     // write-out role: 16-byte chunk (8 channels) r8 of pixel 8 t + rp
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = pass * 2 + jj;
      float4 bb[4];                                          // read before the staging writes (same LDS: the compiler cannot hoist them)
#pragma unroll
      for (int g = 0; g < 4; ++g) bb[g] = *(const float4*)(bsm + wn * 128 + j * 32 + 4 * lhi + 8 * g);
      act_switch(a.act, [&](auto act_c) __attribute__((always_inline)) {     // element loop compiled per activation
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int i = 0; i < XB; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b = bb[g];
            float v[4] = {acc[i][j][4 * g] * a.alpha + b.x, acc[i][j][4 * g + 1] * a.alpha + b.y,
                          acc[i][j][4 * g + 2] * a.alpha + b.z, acc[i][j][4 * g + 3] * a.alpha + b.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ACT);
            *(uint2*)(stg + (i * 32 + l31) * SROW + (jj * 32 + 4 * lhi + 8 * g) * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
          }
      });
    }
    const int cl0 = wn * 128 + pass * 64 + r8 * 8;           // this lane's 8 channels inside the tile
    const bool nok = n0 + cl0 < a.N;
    int64_t orow[8];
    uint4 rres[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int p = 8 * t + rp;
      const int y = y0 + XB * wm + (p >> 5), x = x0 + (p & 31);
      orow[t] = (((int64_t)img * a.H + y) * a.W + x) * a.ldd + n0 + cl0;
      rres[t] = make_uint4(0, 0, 0, 0);
      if (a.R && nok) {
        const int64_t rr = a.res_up ? (((int64_t)img * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1)) * a.ldr
                                    : (((int64_t)img * a.H + y) * a.W + x) * a.ldr;
        rres[t] = *(const uint4*)((const u16*)a.R + rr + n0 + cl0);
      }
    }
    float cs[16];                                           // [0..7] sums, [8..15] sums of squares of the lane's channels
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      uint4 v = *(const uint4*)(stg + (8 * t + rp) * SROW + r8 * 16);
      if (a.R || a.stats) {
        float f[8];
        unpack8<T>(v, f);
        if (a.R) {
          float r[8];
          unpack8<T>(rres[t], r);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += r[e];
          v = pack8<T>(f);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { cs[e] += f[e]; cs[8 + e] += f[e] * f[e]; }
      }
      if (nok) *(uint4*)((u16*)a.D + orow[t]) = v;
    }
    if (a.stats) {
      // butterfly over the 8 lanes that share r8 (lane bits 3..5), halving the live values per step: 14 shuffles
      float b8[8], b4[4], b2[2];
      const bool h3 = lane & 8, h4 = lane & 16, h5 = lane & 32;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float mine = h3 ? cs[k + 8] : cs[k], other = h3 ? cs[k] : cs[k + 8]; b8[k] = mine + __shfl_xor(other, 8); }
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float mine = h4 ? b8[k + 4] : b8[k], other = h4 ? b8[k] : b8[k + 4]; b4[k] = mine + __shfl_xor(other, 16); }
#pragma unroll
      for (int k = 0; k < 2; ++k) { const float mine = h5 ? b4[k + 2] : b4[k], other = h5 ? b4[k] : b4[k + 2]; b2[k] = mine + __shfl_xor(other, 32); }
      // the lane now holds the totals of value index h3 * 8 + h4 * 4 + h5 * 2 + k: statistic h3 of channel h4 * 4 + h5 * 2 + k
      const int e0 = (h4 ? 4 : 0) + (h5 ? 2 : 0);
      float* const slot = stat + wm * 2 * BN;
      slot[2 * (cl0 + e0) + (h3 ? 1 : 0)] = b2[0];
      slot[2 * (cl0 + e0 + 1) + (h3 ? 1 : 0)] = b2[1];
    }
  }
  STAMP(3);
  if (a.stats) {
    __syncthreads();
    float* o = a.stats + (((int64_t)img * a.stats_p + ty * tiles_x + tx) * a.N + n0) * 2;
    for (int c = tid; c < 2 * BN; c += NT) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += stat[w * 2 * BN + c];
      if (n0 + (c >> 1) < a.N) o[c] = v;
    }
  }
  }
#ifdef PMI_STAMPS
  __syncthreads();
  if (tid == 0 && a.ws) {
    ((long long*)a.ws)[(int64_t)blockIdx.x * 8 + 4] = (long long)wall_clock64();
    unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ((long long*)a.ws)[(int64_t)blockIdx.x * 8 + 5] = ((long long)xcc << 32) | hwid;
  }
#endif
}

template <typename T, int PRO>
int launch_p(const pmi_igemm_args& a, hipStream_t s, int cfg) {
  const int nimg = a.M / (a.H * a.W);
  if (cfg == 0) {          // 8x32 px x 256 ch, 8 waves, one workgroup per CU
    const int tiles = nimg * (a.H / 8) * (a.W / 32) * ((a.N + 255) / 256);
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, 4, 2, PRO, PRO == 0, 4, 3>), dim3(tiles), dim3(512), 0, s, a);
  } else if (cfg == 1) {   // 16x32 px x 128 ch, 8 waves
    const int tiles = nimg * (a.H / 16) * (a.W / 32) * ((a.N + 127) / 128);
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, 8, 1, PRO, PRO == 0>), dim3(tiles), dim3(512), 0, s, a);
  } else if (cfg == 3) {   // 8x32 px x 32 ch, 4 waves: convs with a handful of output channels (memory-bound on the input)
    const int tiles = nimg * (a.H / 8) * (a.W / 32) * ((a.N + 31) / 32);
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, 4, 1, PRO, false, 1>), dim3(tiles), dim3(256), 0, s, a);
  } else {                 // 8x32 px x 128 ch, 4 waves, two workgroups per CU overlap each other's staging / epilogue
    const int tiles = nimg * (a.H / 8) * (a.W / 32) * ((a.N + 127) / 128);
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, 4, 1, PRO, false>), dim3(tiles), dim3(256), 0, s, a);
  }
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

template <typename T>
int launch_t(const pmi_igemm_args& a, hipStream_t s, int cfg) {
  if (!a.pro_a) return launch_p<T, 0>(a, s, cfg);
  switch (a.pro_act) {
    case PMI_ACT_NONE: return launch_p<T, 1 + PMI_ACT_NONE>(a, s, cfg);
    case PMI_ACT_RELU: return launch_p<T, 1 + PMI_ACT_RELU>(a, s, cfg);
    case PMI_ACT_SILU: return launch_p<T, 1 + PMI_ACT_SILU>(a, s, cfg);
    default: return PMI_ERR_ARG;
  }
}

}  // namespace

// Returns the config the halo kernel can run (0: 8x32 x 256ch, 1: 16x32 x 128ch) or -1 if the shape needs the generic kernel.
static int g_prefer0 = 1;      // pmi_set_option(2, v): prefer the 8-wave 256-channel config where the grid allows (A/B)
static int g_force_cfg = -1;   // pmi_set_option(1, cfg): force a tile config where eligible (A/B benchmarking)
static int g_wd = 1;           // pmi_set_option(6, v): allow the weights-direct kernel (conv_wd.hip) where fragment-ordered weights are given
static int g_wd128 = 1;        // pmi_set_option(8, v): allow config 7 (128-channel weights-direct tiles)
static int g_wd_mf16 = 1;      // pmi_set_option(7, v): its v_mfma_f32_16x16x32 form (config 6) rather than 32x32x16 (config 4)
void pmi_conv3x3_allow_wd(int v) { g_wd = v; }
void pmi_conv3x3_wd_mf16(int v) { g_wd_mf16 = v; }
void pmi_conv3x3_wd128(int v) { g_wd128 = v; }
static int g_wd_smallc = 1;    // pmi_set_option(13, v): allow config 8 (at most 32 input channels)
void pmi_conv3x3_wd_smallc(int v) { g_wd_smallc = v; }
void pmi_conv3x3_force_config(int cfg) { g_force_cfg = cfg; }
void pmi_conv3x3_prefer_256(int v) { g_prefer0 = v; }

// Returns the tile config the halo kernel runs for this shape (0: 8x32 px x 256 ch / 8 waves, 1: 16x32 x 128 / 8 waves,
// 2: 8x32 x 128 / 4 waves x 2 workgroups per CU, 3: 8x32 px x <= 32 channels) or -1 if the shape needs the generic kernel.
extern "C" int pmi_conv3x3_halo_config(const pmi_igemm_args* a) {
  if (a->taps != 9 || a->stride != 1 || a->batch > 1) return -1;
  const int Cin = a->C0 + a->C1;
  // config 8: at most 32 input channels (the first convolution of the UNets): weights-direct tile with the whole K in registers
  // (precise / mixed mode: the split input's 2 x 8 physical channels are this K too; the output is then split as well, Cout a multiple of 128)
  if (a->Bf && g_wd && g_wd_mf16 && g_wd_smallc && a->C1 == 0 && !a->A1 && a->C0 <= 32 && (a->C0 % 8) == 0 && !a->up && (a->W % 32) == 0 && (a->H % 8) == 0 && !a->pro_a &&
      ((!a->split_out && !a->split_in) || (a->split_out == 32 && a->split_in == 1 && (a->N % 128) == 0 && a->dtype != PMI_DT_BF16)) &&
      !a->out_f32 && !(a->R && a->res_f32) && (a->N % 32) == 0 && a->N >= 64 && (g_force_cfg < 0 || g_force_cfg == 8) &&
      (long)(a->M / (a->H * a->W)) * (a->H / 8) * (a->W / 32) * ((a->N + 127) / 128) >= (g_force_cfg == 8 ? 1 : 256))
    return 8;
  if ((Cin % 64) || (a->C0 % 64) || (a->W % 32) || (a->H % 8)) return -1;
  // (a split input without a prologue whose output is NOT split -- the UNet's last convolution, fp32 out -- is a plain convolution over its
  // 2C physical channels against duplicated weights: it takes the plain route below)
  const bool plain_k = a->split_in == 1 && !a->split_out && !a->pro_a && a->out_f32 && a->N <= 32;
  if ((a->split_out || a->split_in) && !plain_k) {
    // precise / mixed mode (hi + lo f16 pairs): the weights-direct kernel's 16x16x32 configs only.  Without a prologue a split input is an
    // ordinary K dimension of 2C physical channels against duplicated weights; with one the kernel stages (hi, lo) pairs (split_in 1: doubled
    // operand, 2: single operand over the logical channels).  The split epilogue writes [C/32][hi | lo] and takes the output statistics.
    if (!a->Bf || !g_wd || !g_wd_mf16 || a->split_out != 32 || a->out_f32 || (a->N % 128) || a->dtype == PMI_DT_BF16) return -1;      // (an fp32 residual is fine: the same bytes as a split one)
    if (a->split_in == 2 && !a->pro_a) return -1;
    if (a->pro_a && (a->pro_act != PMI_ACT_SILU || !a->split_in)) return -1;
    const int cin_l = a->split_in == 1 ? Cin / 2 : Cin;     // the prologue's coefficient table counts logical channels
    const int t8s = a->M / (a->H * a->W) * (a->H / 8) * (a->W / 32);
    const bool ok6 = (a->N % 256) == 0 && (!a->pro_a || cin_l <= 2048), ok7 = !a->pro_a || cin_l <= 1024;
    if (g_force_cfg == 6 || g_force_cfg == 7) return (g_force_cfg == 6 ? ok6 : ok7) ? g_force_cfg : -1;
    if (g_force_cfg >= 0) return -1;
    if (ok6 && t8s * (a->N / 256) >= 192) return 6;
    if (g_wd128 && ok7 && t8s * (a->N / 128) >= 128) return 7;
    return -1;
  }
  // config 3: at most 32 output channels (the UNet's last conv, 128 -> 6): one MFMA block column per wave
  if (a->N <= 32 && (a->N % 4) == 0 && !a->stats && !a->res_up && a->M / (a->H * a->W) * (a->H / 8) * (a->W / 32) >= 256) return 3;
  if (a->out_f32 || (a->R && a->res_f32)) return -1;
  if (a->N % 128) {
    // Cout a multiple of 32 but not of 128 (StableDiffusion's 320-channel level): the 128-channel weights-direct tiles with a masked tail
    // tile, where the grid still fills the chip and the tail tile is at least half full
    const int t8 = a->M / (a->H * a->W) * (a->H / 8) * (a->W / 32);
    const bool ok7t = a->Bf && g_wd && g_wd_mf16 && (a->N % 32) == 0 && (a->N % 128) >= 64 && a->N > 128 && (!a->pro_a || a->C0 + a->C1 <= 1024);
    if (g_force_cfg == 7) return ok7t ? 7 : -1;
    return (g_force_cfg < 0 && g_wd128 && ok7t && t8 * ((a->N + 127) / 128) >= 256) ? 7 : -1;
  }
  if (a->Bf && g_wd && (!a->pro_a || a->C0 + a->C1 <= 2048)) {   // weights-direct kernel (its prologue coefficient table holds 2048 channels): 4 / 6 = 256-channel tiles (64-channel chunks), 7 = 128-channel tiles (32-channel chunks)
    const int t8 = a->M / (a->H * a->W) * (a->H / 8) * (a->W / 32);
    const bool ok4 = (a->N % 256) == 0;
    if (g_force_cfg == 4 && ok4) return 4;
    if (g_force_cfg == 6 && ok4) return 6;
    if (g_force_cfg < 0 && ok4 && t8 * (a->N / 256) >= 192) return g_wd_mf16 ? 6 : 4;
    // config 7: 128-channel tiles, 4 waves, 32-channel chunks, two workgroups per CU (its coefficient table holds 1024 channels)
    const bool ok7 = g_wd_mf16 && (!a->pro_a || a->C0 + a->C1 <= 1024);
    if (g_force_cfg == 7 && ok7) return 7;
    if (g_force_cfg < 0 && g_wd128 && ok7 && !ok4 && t8 * (a->N / 128) >= (a->N > 512 ? 128 : 256)) return 7;   // (640-channel layers on 32x32 maps at batch 8: 160 tiles)
    // 256-multiple Cout on a map too small for the 256-channel tiles (32x32 at batch 8): the 128-channel tiles give twice the workgroups
    // and measure 5-10 % ahead of the halo kernel's 4-wave config there
    if (g_force_cfg < 0 && g_wd128 && ok7 && ok4 && t8 * (a->N / 128) >= 128) return 7;
  }
  const bool ok0 = (a->N % 256) == 0 || a->N >= 256, ok1 = (a->H % 16) == 0;
  if (g_force_cfg == 0 && ok0) return 0;
  if (g_force_cfg == 1 && ok1) return 1;
  if (g_force_cfg == 2) return 2;
  // One 8-wave workgroup per CU: a grid well below 256 workgroups leaves CUs idle; such layers (<= 32x32 feature maps
  // at batch 8) go to the generic kernel, whose 128x128 tiles (and split-K) fill the chip.
  const int nimg = a->M / (a->H * a->W);
  const int px_tiles8 = nimg * (a->H / 8) * (a->W / 32);
  // Measured per shape (tools/conv_sweep.sh).  Two 4-wave workgroups per CU (cfg 2) hide each workgroup's prologue / epilogue
  // behind the other's main loop: all workgroups of a launch run in lockstep, so with one workgroup per CU every epilogue hits
  // HBM at the same moment.  With 256-channel tiles one 8-wave workgroup per CU (cfg 0: the patch is staged once for 256
  // output channels) is 3-12 % ahead again since the weight stores moved under the MFMAs.
  if (g_prefer0 && (a->N % 256) == 0 && px_tiles8 * (a->N / 256) >= 256) return 0;
  if (px_tiles8 * (a->N / 128) >= 128) return 2;      // even at one workgroup per two CUs the fused prologue / statistics beat the generic path
  if ((a->N % 256) == 0) return px_tiles8 * (a->N / 256) >= 192 ? 0 : -1;
  if (ok1 && (px_tiles8 / 2) * (a->N / 128) >= 192) return 1;
  return -1;
}

int pmi_conv3x3_halo_launch(const pmi_igemm_args* a, int cfg, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  return a->dtype == PMI_DT_BF16 ? launch_t<BF16>(*a, s, cfg) : launch_t<F16>(*a, s, cfg);
}
