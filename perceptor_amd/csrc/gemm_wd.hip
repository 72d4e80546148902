// Plain GEMM for gfx950 MFMA in the "weights-direct" form of conv_wd.hip -- the ViT linears of the CLIP tower, their
// input-gradient GEMMs (dX = dY W) and the single-source 1x1 convolutions / Conv1d k=1 of the UNets.
//
//   D[m][n] = act(alpha * sum_k A[m][k] * B[n][k] + bias[n]) + R[m][n]
//
// A 512-thread workgroup owns a (16 MB) x 256 tile (MB = 8 or 9: M = 8 x 257 = 2056 rows of the ViT-L/14 batch are 15 tiles of
// 144 rows -> 240 workgroups for N = 4096, one round on 256 CUs; 128-row tiles would need 17 x 16 = 272).  Every wave owns
// 16 MB rows x 32 columns (2 MB blocks of v_mfma_f32_16x16x32: 8 MB accumulator registers) and streams the weights of ITS 32
// columns global -> VGPR in MFMA fragment order (host-packed: engine/ops.py PackedLinear.frag_gemm), prefetched three 32-deep
// k-steps ahead in a 4-slot register ring: no LDS, no synchronisation on the weight side.  The activation tile goes through
// LDS, 128 k at a time, double-buffered: ONE barrier per 128-deep chunk (8 MB MFMAs per wave between barriers).  Fragment
// registers are reloaded in place: block mb's register is dead after its two MFMAs and takes the NEXT step's block mb at once
// (the reload has (MB - 1) x 32 cycles of MFMAs to land under).  LDS row pitch 288 B = 18 slots of 16 B: the 16 lanes of a
// ds_read_b128 group (rows i at slot 18 i + k-quarter) hit 16 different slots.
// Epilogue: two passes (columns 0..127, 128..255) through an fp32 LDS image, written out as whole rows (512 B fp32 / 256 B
// 16-bit per row): bias, activation, residual (fp32 or 16-bit), fp32 or 16-bit output; with split-K (grid.z) raw fp32 slabs
// for splitk_reduce_kernel (igemm.hip).
#include "common.h"
#include "../../include/perceptor_hip.h"

namespace {

// TWO: the K dimension is the concatenation of two activation tensors (A0: C0 channels, A1: C1; C0 a multiple of 128) -- the 1x1
// `skip_connection` of a ResBlock whose input is a skip concat (th.cat([h, hs.pop()]), unet.py:647-650), never materialised.
// NWV: waves per workgroup = 32-column slices per tile: 8 (256-column tiles) or 4 (128-column tiles, 256 threads: N = 128, where a 256-column
// tile would run half its waves on zeros).
// CONV: stride-1 3x3 convolution as the same GEMM (K = 9 taps x Cin, k = tap * Cin + c as PackedLinear packs it; Cin of every source a
// multiple of 128, so a 128-deep chunk has one tap and one source): the activation row of output pixel m for tap (dy, dx) is the input
// row of pixel (y + dy - 1, x + dx - 1) -- the staging loads shift their row offset per chunk and read zeros outside the image.  For the
// maps the conv3x3 kernels' 8 x 32-pixel tiles do not fit or fill (16x16, 8x8, 4x4 at batch 8), with split-K like any long-K GEMM.
template <typename T, int MB, bool TWO = false, int NWV = 8, bool CONV = false>
__global__ __launch_bounds__(NWV * 64, 2) void gemm_wd_kernel(const pmi_igemm_args a) {
  constexpr int TM = 16 * MB, BN = NWV * 32, BK = 128, NT = NWV * 64;
  constexpr int ROW = 2 * BK + 32;                     // LDS row pitch (bytes)
  constexpr int TILE = TM * ROW;
  constexpr int NPI = (TM * 16 + NT - 1) / NT;         // 16-byte staging pieces per thread per chunk
  constexpr int HIMG = TM * (BN * 2 + 16);             // one 16-bit epilogue image of the tile
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE > 2 * HIMG ? 2 * TILE : 2 * HIMG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PMI_STAMPS   // tools/gemm_probe.py --stamps: phase timestamps (100 MHz) per workgroup; a.reserved carries the enable flag
#define GSTAMP(k) do { if (tid == 0 && a.reserved == 77 && a.splitk <= 1) ((long long*)a.ws)[(int64_t)blockIdx.x * 8 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define GSTAMP(k) do {} while (0)
#endif
  GSTAMP(0);
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + TM - 1) / TM;   // N a multiple of 32: the last tile's waves past N multiply zeros and store nothing
  const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  // every XCD (own 4 MB L2) gets a contiguous range of tiles: let it walk ALL tiles of the smaller operand for a few of the larger
  const bool mfast = (int64_t)a.M <= (int64_t)a.N;
  const int tm = mfast ? logical % tiles_m : logical / tiles_n, tn = mfast ? logical / tiles_m : logical % tiles_n;
  const int m0 = tm * TM, n0 = tn * BN;
  const int nch_all = (a.K + BK - 1) / BK;          // the packed weights are zero-padded to whole 128-deep chunks
  int c0 = 0, nch = nch_all;
  if (a.splitk > 1) {
    const int per = (nch_all + a.splitk - 1) / a.splitk;
    c0 = blockIdx.z * per;
    nch = min(per, nch_all - c0);
  }

  // activation rows of this tile: rows past M fall outside the resource (zeros)
  const int rows = min(TM, a.M - m0);
  const int K0 = TWO ? a.C0 : a.K;                      // channels of the first source
  // (CONV: the resources cover the whole tensors -- a tap reaches rows of the neighbouring tiles; the host checks they stay below 2 GB)
  const int64_t mb_ = CONV ? 0 : m0, mr_ = CONV ? a.M : rows;
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc((const u16*)a.A0 + mb_ * a.lda0, ((mr_ - 1) * a.lda0 + K0) * 2);
  const __amdgpu_buffer_rsrc_t rsrc_a1 = TWO ? make_rsrc((const u16*)a.A1 + mb_ * a.lda1, ((mr_ - 1) * a.lda1 + a.C1) * 2) : rsrc_a;
  uint32_t pvo[NPI];                                    // byte offset of this thread's staging pieces inside the tile's rows (k = 0)
  uint32_t pvo1[TWO ? NPI : 1];                         // the same for the second source's row pitch
  uint32_t pyx[CONV ? NPI : 1];                         // CONV: (y << 16) | x of the piece's output pixel; rows past M: far outside every image
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int p = tid + NT * i, r = p >> 4, c16 = p & 15;
    const int rg = CONV ? m0 + r : r;                   // CONV: offsets from the tensor's first row
    pvo[i] = r < rows ? (uint32_t)(rg * a.lda0 + c16 * 8) * 2u : PMI_BUF_OOB;
    if constexpr (TWO) pvo1[i] = r < rows ? (uint32_t)(rg * a.lda1 + c16 * 8) * 2u : PMI_BUF_OOB;
    if constexpr (CONV) {
      const int hwm = rg % (a.H * a.W), y = hwm / a.W;
      pyx[i] = r < rows ? ((uint32_t)y << 16) | (uint32_t)(hwm - y * a.W) : 0x40004000u;
    }
  }
  // this wave's weight stream: [chunk][k32 (4)][16-column block (2)][lane][8], 8 KB per chunk, contiguous
  const int64_t wslab = (int64_t)nch_all * 8192;
  const bool wave_live = n0 + wid * 32 < a.N;          // wave-uniform; a dead wave's weight resource is empty (every load returns zeros)
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc((const char*)a.Bf + (wave_live ? (int64_t)(tn * NWV + wid) * wslab : 0), wave_live ? wslab : 0);
  const uint32_t wvo = (uint32_t)lane * 16u + (uint32_t)c0 * 8192u;

  uint4 pr[NPI];
  const int kc16 = (tid & 15) * 8;                      // first channel of this thread's staging pieces inside a 128-deep chunk (NT is a multiple of 16)
  auto load_tile = [&](int chunk, bool live) {
    if constexpr (CONV) {
      const int cin = a.C0 + a.C1;
      const int k0 = (c0 + chunk) * BK;                 // wave-uniform: tap and source of this chunk
      const int tap = __builtin_amdgcn_readfirstlane(k0 / cin), kc = k0 - tap * cin;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const bool second = TWO && kc >= a.C0;
      const __amdgpu_buffer_rsrc_t rs = second ? rsrc_a1 : rsrc_a;
      const uint32_t so = (uint32_t)(second ? kc - a.C0 : kc) * 2u;
      const int shift = (dy * a.W + dx) * (second ? a.lda1 : a.lda0) * 2;      // bytes from the output pixel's row to the tap's row
#pragma unroll
      for (int i = 0; i < NPI; ++i) {
        const int yy = (int)(pyx[i] >> 16) + dy, xx = (int)(pyx[i] & 0xffffu) + dx;
        const bool in = live && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;     // zero padding, and rows past M
        const uint32_t pv = TWO ? (second ? pvo1[i] : pvo[i]) : pvo[i];
        pr[i] = buf_load16(rs, in ? pv + (uint32_t)shift : PMI_BUF_OOB, so);
      }
    } else if constexpr (TWO) {
      const int k0 = (c0 + chunk) * BK;                 // wave-uniform: which tensor this 128-deep chunk comes from (C0 is a multiple of 128)
      const bool second = k0 >= a.C0;
      const __amdgpu_buffer_rsrc_t rs = second ? rsrc_a1 : rsrc_a;
      const uint32_t so = (uint32_t)(second ? k0 - a.C0 : k0) * 2u;
      // K tail (C1 a multiple of 32 only): pieces past the source's last channel must read zeros, not the next row -- a NaN / Inf there
      // would survive the multiplication by the zero-padded weights
      const bool cut = (second ? k0 - a.C0 + kc16 >= a.C1 : false);
#pragma unroll
      for (int i = 0; i < NPI; ++i) pr[i] = buf_load16(rs, (live && !cut) ? (second ? pvo1[i] : pvo[i]) : PMI_BUF_OOB, so);
    } else {
      const bool cut = (c0 + chunk) * BK + kc16 >= a.K;         // K tail, as above (only the last chunk of a K that is not a multiple of 128)
#pragma unroll
      for (int i = 0; i < NPI; ++i) pr[i] = buf_load16(rsrc_a, (live && !cut) ? pvo[i] : PMI_BUF_OOB, (uint32_t)(c0 + chunk) * (BK * 2));
    }
  };
  auto store_tile = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int p = tid + NT * i;
      if (NPI * NT == TM * 16 || p < TM * 16) *(uint4*)(buf + (p >> 4) * ROW + (p & 15) * 16) = pr[i];
    }
  };

  f32x4 acc[MB][2];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 wq[4][2];                                       // [k32-step slot][16-column block]
  auto load_w = [&](int slot, int step) {               // global step index inside this workgroup's k range; past the end: zeros
    const uint32_t vo = wvo + (uint32_t)step * 2048u;
    const bool live = step < nch * 4;
    wq[slot][0] = buf_load16(rsrc_w, live ? vo : PMI_BUF_OOB, 0);
    wq[slot][1] = buf_load16(rsrc_w, live ? vo + 1024u : PMI_BUF_OOB, 0);
  };

  load_w(0, 0); load_w(1, 1); load_w(2, 2);
  load_tile(0, true);
  store_tile(smem);
  load_tile(nch > 1 ? 1 : 0, nch > 1);                 // chunk 1 travels while chunk 0 is multiplied
  __syncthreads();
  GSTAMP(1);
  const int frag0 = (lane & 15) * ROW + (lane >> 4) * 16;
  uint4 xf[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) xf[mb] = *(const uint4*)(smem + frag0 + mb * 16 * ROW);

  for (int chunk = 0; chunk < nch; ++chunk) {
    const char* const pb = smem + (chunk & 1) * TILE;
    char* const pn = smem + ((chunk + 1) & 1) * TILE;
    const bool more2 = chunk + 2 < nch;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      load_w((s + 3) & 3, chunk * 4 + s + 3);
      if (s == 2) store_tile(pn);                       // chunk + 1 (loaded a whole chunk ago) -> the other buffer
      if (s == 3) {
        load_tile(more2 ? chunk + 2 : chunk, more2);    // the staging registers are free again: chunk + 2 starts its trip
        __syncthreads();                                // next tile complete; every wave has issued its last read of the buffer it overwrites next
      }
      // (round 3: storing at s == 0 and reloading at once -- a full chunk of latency cover instead of three quarters -- measured no change:
      //  2056 x 4096 x 1024 23.5-24.2 vs 24.3-25.2 us, c5 step 42.2-42.3 vs 42.1-42.2 ms; the loop, 1.9 us per 128-deep chunk against 0.6 us
      //  of MFMA work, waits on the per-wave weight stream as much as on the activation tile)
      __builtin_amdgcn_sched_barrier(0);
      const char* const nb = (s < 3 ? pb + (s + 1) * 64 : pn) + frag0;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        acc[mb][0] = T::mfma16(wq[s][0], xf[mb], acc[mb][0]);
        acc[mb][1] = T::mfma16(wq[s][1], xf[mb], acc[mb][1]);
        xf[mb] = *(const uint4*)(nb + mb * 16 * ROW);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                      // all fragment reads done: the buffers become the epilogue image
  GSTAMP(2);

  const bool raw = a.splitk > 1;
  if (a.act == PMI_ACT_GEGLU) {
    // ---- GEGLU epilogue (stable_diffusion/attention.py:346-348): the weights are packed so that every wave's 32 columns are 16 value
    // columns (block 0) and their 16 gate columns (block 1): out = value * gelu(gate), 128 output columns per tile, through a 16-bit
    // LDS image and out as 256-byte rows.  The 8C-wide projection never exists in HBM. ----
    constexpr int GROW = (BN / 2) * 2 + 16;
    const int cl = wid * 16 + 4 * (lane >> 4);           // output column inside the tile
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv;
    if (a.bias && wave_live) { bv = *(const float4*)(a.bias + n0 + wid * 32 + 4 * (lane >> 4)); bg = *(const float4*)(a.bias + n0 + wid * 32 + 16 + 4 * (lane >> 4)); }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const f32x4 cv = acc[mb][0], cg = acc[mb][1];
      float v[4] = {cv[0] * a.alpha + bv.x, cv[1] * a.alpha + bv.y, cv[2] * a.alpha + bv.z, cv[3] * a.alpha + bv.w};
      const float g[4] = {cg[0] * a.alpha + bg.x, cg[1] * a.alpha + bg.y, cg[2] * a.alpha + bg.z, cg[3] * a.alpha + bg.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= act_apply(g[e], PMI_ACT_GELU);
      *(uint2*)(smem + (mb * 16 + (lane & 15)) * GROW + cl * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    const int gc8 = tid % (BN / 16), grow = tid / (BN / 16);     // 8 columns (16 B) x rows grow + 32 j
    const int nh = a.N >> 1, nh0 = n0 >> 1;
#pragma unroll
    for (int j = 0; j < (TM + 31) / 32; ++j) {
      const int r = grow + 32 * j, m = m0 + r;
      if (r >= TM || m >= a.M || nh0 + gc8 * 8 >= nh) continue;
      *(uint4*)((u16*)a.D + (int64_t)m * a.ldd + nh0 + gc8 * 8) = *(const uint4*)(smem + r * GROW + gc8 * 16);
    }
    GSTAMP(3);
    return;
  }
  if (!raw && !a.out_f32 && !(a.R && a.res_f32)) {
    // ---- epilogue, 16-bit output: ONE pass through a 16-bit image of the whole tile (bias + activation applied on the way in),
    // written out as 512-byte rows, 16 bytes per lane; a 16-bit residual is added on the way out ----
    constexpr int HROW = BN * 2 + 16;
    static_assert(2 * TM * HROW <= 160 * 1024, "the two 16-bit epilogue images (output, pre-activation) must fit the LDS");
    act_switch(a.act, [&](auto act_c) __attribute__((always_inline)) {    // element loops compiled per activation (no per-value switch)
      constexpr int ACT = decltype(act_c)::value;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int nl = wid * 32 + cb * 16 + 4 * (lane >> 4);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias && wave_live) bv = *(const float4*)(a.bias + n0 + nl);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const f32x4 c = acc[mb][cb];
          float v[4] = {c[0] * a.alpha + bv.x, c[1] * a.alpha + bv.y, c[2] * a.alpha + bv.z, c[3] * a.alpha + bv.w};
          if (ACT != PMI_ACT_NONE && a.D2)
            *(uint2*)(smem + TM * HROW + (mb * 16 + (lane & 15)) * HROW + nl * 2) = pack4<T>(v[0], v[1], v[2], v[3]);   // pre-activation image
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ACT);
          *(uint2*)(smem + (mb * 16 + (lane & 15)) * HROW + nl * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
        }
      }
    });
    __syncthreads();
    const bool d2_img = a.D2 && a.act != PMI_ACT_NONE;   // without an activation the second output equals the first
    const int pc8 = tid % (BN / 8), prow = tid / (BN / 8);     // 8 columns (16 B) x rows prow + 16 j
    // residual / activation-gradient operand of all MB rows of this thread, loaded up front: read inside the loop below they sit behind the
    // previous row's store (the compiler must assume the tensors alias), one exposed memory latency per row
    uint4 rv[MB], av[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const int m = m0 + prow + 16 * j;
      const bool live = m < a.M && n0 + pc8 * 8 < a.N;
      rv[j] = av[j] = make_uint4(0, 0, 0, 0);
      if (live && a.aux) av[j] = *(const uint4*)((const u16*)a.aux + (int64_t)m * a.ldd + n0 + pc8 * 8);
      if (live && a.R) rv[j] = *(const uint4*)((const u16*)a.R + (int64_t)m * a.ldr + n0 + pc8 * 8);
    }
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const int r = prow + 16 * j, m = m0 + r;
      if (m >= a.M || n0 + pc8 * 8 >= a.N) continue;
      uint4 v = *(const uint4*)(smem + r * HROW + pc8 * 16);
      if (a.R || a.aux) {
        float f[8], rr[8];
        unpack8<T>(v, f);
        if (a.aux) {
          unpack8<T>(av[j], rr);
          switch (a.aux_act) {                            // the activation code is folded per case (no per-value switch)
            case PMI_ACT_GELU:
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] *= act_grad(rr[e], PMI_ACT_GELU);
              break;
            case PMI_ACT_QUICKGELU:
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] *= act_grad(rr[e], PMI_ACT_QUICKGELU);
              break;
            default:
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] *= act_grad(rr[e], a.aux_act);
          }
        }
        if (a.R) {
          unpack8<T>(rv[j], rr);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += rr[e];
        }
        v = pack8<T>(f);
      }
      *(uint4*)((u16*)a.D + (int64_t)m * a.ldd + n0 + pc8 * 8) = v;
      if (a.D2) {
        const uint4 v2 = *(const uint4*)(smem + (d2_img ? TM * HROW : 0) + r * HROW + pc8 * 16);
        *(uint4*)((u16*)a.D2 + (int64_t)m * a.ldd + n0 + pc8 * 8) = v2;
      }
    }
    GSTAMP(3);
    return;
  }
  // ---- epilogue, fp32 output / fp32 residual / split-K slabs: straight from the accumulators -- a lane holds 4 consecutive columns
  // (16 bytes fp32) of one row, the four quarter-waves of a block complete 64-byte runs, the two blocks of a wave a 128-byte line ----
  float* const Dslab = raw ? (float*)a.ws + (int64_t)blockIdx.z * a.M * a.N : nullptr;
  act_switch(raw ? PMI_ACT_NONE : a.act, [&](auto act_c) __attribute__((always_inline)) {
  constexpr int ACT = decltype(act_c)::value;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int n = n0 + wid * 32 + cb * 16 + 4 * (lane >> 4);
    if (!wave_live) continue;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!raw && a.bias) bv = *(const float4*)(a.bias + n);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = m0 + mb * 16 + (lane & 15);
      if (m >= a.M) continue;
      const f32x4 c = acc[mb][cb];
      if (raw) {
        *(float4*)(Dslab + (int64_t)m * a.N + n) = make_float4(c[0], c[1], c[2], c[3]);
        continue;
      }
      float v[4] = {c[0] * a.alpha + bv.x, c[1] * a.alpha + bv.y, c[2] * a.alpha + bv.z, c[3] * a.alpha + bv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ACT);
      if (a.R) {
        if (a.res_f32) {
          const float4 rr = *(const float4*)((const float*)a.R + (int64_t)m * a.ldr + n);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        } else {
          const uint2 rr = *(const uint2*)((const u16*)a.R + (int64_t)m * a.ldr + n);
          v[0] += T::to_f((u16)(rr.x & 0xffff)); v[1] += T::to_f((u16)(rr.x >> 16));
          v[2] += T::to_f((u16)(rr.y & 0xffff)); v[3] += T::to_f((u16)(rr.y >> 16));
        }
      }
      if (a.out_f32) *(float4*)((float*)a.D + (int64_t)m * a.ldd + n) = make_float4(v[0], v[1], v[2], v[3]);
      else *(uint2*)((u16*)a.D + (int64_t)m * a.ldd + n) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
  }
  });
  GSTAMP(3);
}

static int g_few_wgs = 128;        // A/B switch: pmi_set_option(12, n): 256-column grids below n workgroups use the 128-column tiles

template <typename T>
int launch(const pmi_igemm_args& a, hipStream_t s, int mb) {
  const int tm = 16 * mb;
  // fewer than 128 workgroups of 256 columns and no split-K (short K): the 128-column tiles double the grid (2056 x 1024 x 1024: 14.2 vs 16.7 us)
  const bool few = a.splitk <= 1 && (long)((a.M + 127) / 128) * ((a.N + 255) / 256) < g_few_wgs && !a.D2 && !a.aux && a.act != PMI_ACT_GEGLU;
  const bool half_tail = (a.N % 256) != 0 && (a.N % 256) <= 128 && a.N < 1024;     // e.g. N = 320: 3 tiles of 128 instead of 2 of 256 (one a quarter full)
  if (a.N < 256 || few || half_tail) {                 // 128-column tiles, four waves, two workgroups per CU
    const dim3 g4(((a.M + 127) / 128) * ((a.N + 127) / 128), 1, a.splitk > 1 ? a.splitk : 1);
    if (a.taps == 9) {
      if (a.A1) hipLaunchKernelGGL((gemm_wd_kernel<T, 8, true, 4, true>), g4, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_wd_kernel<T, 8, false, 4, true>), g4, dim3(256), 0, s, a);
    } else if (a.A1) hipLaunchKernelGGL((gemm_wd_kernel<T, 8, true, 4>), g4, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_wd_kernel<T, 8, false, 4>), g4, dim3(256), 0, s, a);
    PMI_CHECK_LAUNCH();
    return PMI_OK;
  }
  const dim3 grid(((a.M + tm - 1) / tm) * ((a.N + 255) / 256), 1, a.splitk > 1 ? a.splitk : 1);
  if (a.taps == 9) {                                   // 3x3 convolution on small maps: 128-row tiles
    if (a.A1) hipLaunchKernelGGL((gemm_wd_kernel<T, 8, true, 8, true>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm_wd_kernel<T, 8, false, 8, true>), grid, dim3(512), 0, s, a);
  } else if (a.A1) hipLaunchKernelGGL((gemm_wd_kernel<T, 8, true>), grid, dim3(512), 0, s, a);        // two-source K: 128-row tiles
  else if (mb == 9) hipLaunchKernelGGL((gemm_wd_kernel<T, 9>), grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL((gemm_wd_kernel<T, 8>), grid, dim3(512), 0, s, a);
  PMI_CHECK_LAUNCH();
  return PMI_OK;
}

}  // namespace

void pmi_gemm_wd_few_wgs(int v) { g_few_wgs = v; }

// rows per tile (128 or 144): fewest rounds of 256 workgroups, then least work per workgroup
int pmi_gemm_wd_tile_rows(const pmi_igemm_args* a, int splitk) {
  if (a->A1 || a->taps == 9 || a->N < 256 || ((a->N % 256) != 0 && (a->N % 256) <= 128 && a->N < 1024)) return 128;
  if (splitk <= 1 && (long)((a->M + 127) / 128) * ((a->N + 255) / 256) < g_few_wgs && !a->D2 && !a->aux && a->act != PMI_ACT_GEGLU) return 128;
  int best = 8;
  long best_cost = -1;
  for (int mb = 8; mb <= 9; ++mb) {
    const long wgs = (long)((a->M + 16 * mb - 1) / (16 * mb)) * ((a->N + 255) / 256) * (splitk > 1 ? splitk : 1);
    const long cost = ((wgs + 255) / 256) * mb;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mb; }
  }
  return 16 * best;
}

// 1 when the weights-direct GEMM takes this call (plain GEMM with fragment-ordered weights), else the generic kernel runs
int pmi_igemm_halo_allowed(void);     // igemm.hip

extern "C" int pmi_gemm_wd_eligible(const pmi_igemm_args* a) {
  if (!a->Bf || (a->taps != 1 && a->taps != 9) || a->up || a->stride != 1 || a->batch > 1) return 0;
  if (a->taps == 9) {
    // 3x3 convolution the conv3x3 kernels do not take (small or odd-width maps): whole 128-deep chunks per tap and source, 32-bit offsets
    if ((a->C0 % 128) || (a->A1 ? (a->C1 <= 0 || (a->C1 % 128)) : a->C1 != 0) || a->D2 || a->aux || a->act == PMI_ACT_GEGLU || a->split_in) return 0;
    if (a->H <= 0 || a->W <= 0 || a->H >= 0x4000 || a->W >= 0x4000 || a->Hin != a->H || a->Win != a->W || (a->M % (a->H * a->W))) return 0;
    if (a->K != 9 * (a->C0 + a->C1) || (int64_t)a->M * (a->lda0 > a->lda1 ? a->lda0 : a->lda1) * 2 >= ((int64_t)1 << 31)) return 0;
    if (pmi_igemm_halo_allowed() && pmi_conv3x3_halo_config(a) >= 0) return 0;
    // below 512 output pixels in all (4x4 maps at batch 4, 8x8 at batch 1..4) a 128-row tile is mostly padding and the generic kernel's
    // 16-way split is as fast or faster (tools/gemm_trace.py --config c2: M = 64: 33 vs 31 us, M = 256: equal, M = 1024: 29.5 vs 33.4)
    if (a->M < 512) return 0;
  } else {
    if (a->A1 ? (a->C1 <= 0 || (a->C0 % 128) || (a->C1 % 32) || a->splitk > 1 || a->D2 || a->aux || a->act == PMI_ACT_GEGLU) : a->C1 != 0) return 0;
    if (a->K != a->C0 + a->C1) return 0;
  }
  // per-sample bias (the timestep projection added behind a ResBlock's first convolution): conv mode only, and only with split-K -- the
  // reduce kernel adds it; splitk == 0 is the caller's query before it has chosen the split (pmi_igemm re-checks with the final value)
  if (a->nbias && (a->taps != 9 || a->splitk == 1)) return 0;
  if (a->res_up && (a->taps != 9 || a->splitk == 1)) return 0;      // (an up-sampled residual: likewise the reduce kernel's)
  if (a->stats || a->pro_a || a->split_out) return 0;
  if ((a->K % 32) || (a->N % 32) || a->M < 64) return 0;      // K tail: zero-padded weights; N tail: masked waves
  if (a->N < 256 && a->N != 128) return 0;  // N = 128: the four-wave 128-column tiles; other narrow matrices stay on the generic kernel
  if ((a->N % 256) && a->N < 1024) {        // narrow matrix with a partly filled last tile: measured per shape against the generic 128-wide tiles (tools/sd_trace.py)
    const int tail = a->N % 256;
    if (tail < 64) return 0;      // (N = 320 runs on the 128-column tiles: see launch())
  }
  if ((a->D2 || a->aux) && (a->out_f32 || a->splitk > 1 || (a->R && a->res_f32))) return 0;
  if (a->act == PMI_ACT_GEGLU && (a->R || a->D2 || a->aux || a->out_f32 || a->splitk > 1)) return 0;
  return 1;
}

int pmi_gemm_wd_launch(const pmi_igemm_args* a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int mb = pmi_gemm_wd_tile_rows(a, a->splitk) / 16;
  return a->dtype == PMI_DT_BF16 ? launch<BF16>(*a, s, mb) : launch<F16>(*a, s, mb);
}
