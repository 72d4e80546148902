// conv3x3 (stride 1, pad 1) for gfx950 MFMA, "weights-direct" form -- the dominant kernel of the UNets.
//
// Every wave owns a 256-pixel x 32-output-channel tile (8 image rows x 32 pixels; 8 MFMA 32x32x16 blocks = 128 accumulator
// registers) and streams ITS OWN weights global -> VGPR, already in MFMA fragment order (packed once on the host, see
// engine/ops.py::pack_frag): one coalesced 1 KB buffer load per fragment, no LDS and no cross-wave synchronisation on the
// weight side.  Only the activation patch goes through LDS: per CK-channel input chunk the (TH+2) x 34 halo patch is
// staged once (double-buffered) and serves all 9 taps of all waves of the workgroup -> ONE workgroup barrier per chunk
// (9 x CK/16 x 8 = 288 (CK = 64) or 144 (CK = 32) MFMAs per wave between barriers; the round-1 kernel had one per tap).
//
// Loop order inside a chunk is (dx, 16-channel k-step) with the three dy taps innermost: the 8 output rows of a wave need the
// same 10 patch-row fragments for dy = 0, 1, 2, so a group of 24 MFMAs reads 10 fragments from LDS (0.42 ds_read_b128 per
// MFMA; round 1: 0.75) and 3 weight fragments from global (prefetched two groups ahead in a 3-slot register ring).
//   NWN: waves along output channels (BN = 32 NWN), NWM: waves along image rows (TH = 8 NWM)
//   <8,1,CK=64>: 8x32 px x 256 ch, one workgroup per CU;  <4,1,CK=32>: 8x32 px x 128 ch, two workgroups per CU
// LDS patch rows hold CK channels; the 16-byte chunk index is XOR-swizzled with the pixel index so that the 16 lanes of a
// ds_read_b128 group (16 consecutive pixels, any tap shift) hit 16 different 16-byte slots.
// Fused: nearest-x2 upsample of the input (patch gather), skip-concat (two sources), GroupNorm-apply(+FiLM)+activation on
// the patch as it is written to LDS (zero padding stays zero); bias / per-sample bias / activation / residual (optionally
// through a nearest-x2 upsample) / per-channel (sum, sumsq) of the output for the next GroupNorm in the epilogue, which
// transposes the whole tile through LDS and writes full pixel rows (BN x 2 bytes contiguous).
#include "common.h"
#include "../../include/perceptor_hip.h"

#ifndef WD_ILV
#define WD_ILV 4     // VALU instructions of the patch staging placed behind each MFMA (0: staging first, then the MFMAs)
#endif

namespace {

constexpr int PW = 34;

// LDS patch layout: one row per patch pixel, CK channels + 16 bytes of padding, NOT swizzled: a fragment read is then
// (per-lane base) + (compile-time offset), one address register for all 10 x 3 x CK/16 reads of a chunk (an XOR swizzle needs a
// lane-dependent address per read; the compiler hoisted those ~120 addresses out of the chunk loop and spilled them).  With a row
// pitch of 144 B (CK = 64) or 80 B (CK = 32) the 16 lanes of a ds_read_b128 group -- 16 consecutive pixels -- start at dwords
// 36 i or 20 i (mod 64), i.e. on 16 different 4-dword slots: conflict-free for every tap shift.

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <typename T, int NWN, int NWM, int CK, int PRO>
__global__ __launch_bounds__(NWN * NWM * 64, 2) void conv3x3_wd_kernel(const pmi_igemm_args a) {
  constexpr int NW = NWN * NWM, NT = NW * 64;
  constexpr int TH = 8 * NWM, PP = (TH + 2) * PW;
  constexpr int CPR = CK / 8;                          // 16-byte chunks per patch pixel
  constexpr int KS = CK / 16;                          // 16-channel k-steps per chunk
  constexpr int NG = 3 * KS;                           // (dx, k-step) groups per chunk, 24 MFMAs each
  constexpr int NPI = (PP * CPR + NT - 1) / NT;        // 16-byte staging pieces per thread per chunk
  constexpr int ROW = CK * 2 + 16;                     // patch row pitch in bytes
  constexpr int PATCH_BYTES = (NPI * NT / CPR) * ROW;  // padded to whole staging passes: no bounds test on the LDS writes
  constexpr int BN = NWN * 32;
  constexpr int NPX = 256 * NWM;
  constexpr int SROW = BN * 2 + 16;                    // epilogue staging row: BN 16-bit channels + 16 B pad
  constexpr int EPI_BYTES = NPX * SROW + NW * BN * 8 + BN * 4;
  constexpr int GB = 3 * 1024;                         // bytes of one group's weight fragments (3 dy x 64 lanes x 16 B)
  static_assert(NPI <= NG, "one staging piece per group");
  __shared__ __attribute__((aligned(16))) char smem[cmax(2 * PATCH_BYTES, EPI_BYTES)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wid / NWN, wn = wid % NWN;

  const int tiles_x = a.W / 32, tiles_y = a.H / TH, tiles_n = a.N / BN;
  const int nimg = a.M / (a.H * a.W);
  int logical = xcd_remap(blockIdx.x, nimg * tiles_y * tiles_x * tiles_n);
  const int tn = logical % tiles_n; logical /= tiles_n;
  const int tx = logical % tiles_x; logical /= tiles_x;
  const int ty = logical % tiles_y;
  const int img = logical / tiles_y;
  const int y0 = ty * TH, x0 = tx * 32, n0 = tn * BN;

#ifdef PMI_STAMPS   // tools/conv_probe.py --stamps: phase timestamps (100 MHz) per workgroup
#define STAMP(k) do { if (tid == 0 && a.ws) ((long long*)a.ws)[(int64_t)blockIdx.x * 8 + (k)] = (long long)wall_clock64(); } while (0)
#define CSTAMP(k) do { if (lane == 0 && a.ws) { ((long long*)a.ws)[(1 << 19) + ((int64_t)blockIdx.x * 8 + wid) * 4 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
                                                ((long long*)a.ws)[(1 << 19) + ((int64_t)blockIdx.x * 8 + wid) * 4 + (k) + 2] = (long long)wall_clock64(); } } while (0)
#else
#define STAMP(k) do {} while (0)
#define CSTAMP(k) do {} while (0)
#endif
  STAMP(0);
  const int Cin = a.C0 + a.C1;
  const int nchunks = Cin / CK;
  const int sc = tid % CPR;                            // 16-byte chunk (8 channels) of a patch pixel this thread stages (NT % CPR == 0)
  const int64_t img_px = (int64_t)a.Hin * a.Win;
  const u16* const A0i = (const u16*)a.A0 + (int64_t)img * img_px * a.lda0;
  const u16* const A1i = a.A1 ? (const u16*)a.A1 + (int64_t)img * img_px * a.lda1 : A0i;
  const int64_t bytes0 = ((img_px - 1) * a.lda0 + a.C0) * 2;
  const int64_t bytes1 = a.A1 ? ((img_px - 1) * a.lda1 + a.C1) * 2 : 0;
  // this wave's weight stream: [chunk][dx][k-step][dy][lane][8] 16-bit, contiguous
  const int64_t wslab = (int64_t)nchunks * NG * GB;
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc((const char*)a.Bf + (int64_t)(tn * NWN + wn) * wslab, wslab);
  const uint32_t wvo = (uint32_t)lane * 16u;

  // ---- patch staging plan: source pixel (inside the image) per staged piece, -1 = zero padding ----
  int ppix[NPI];
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int pp = tid / CPR + (NT / CPR) * i;
    const int py = pp / PW, px = pp - py * PW;
    const int sy = y0 - 1 + py, sx = x0 - 1 + px;
    const bool inside = pp < PP && (unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W;
    ppix[i] = inside ? (sy >> a.up) * a.Win + (sx >> a.up) : -1;
  }

  float ga[8], gb[8];
  // source of chunk `chunk`: which tensor, row pitch, channel offset (wave-uniform: C0 is a multiple of CK)
  auto load_piece = [&](int chunk, int i, bool live) -> uint4 {          // !live: out-of-range offset, zeros, no traffic, no branch
    const int cbase = chunk * CK;
    const bool second = cbase >= a.C0;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(second ? A1i : A0i, second ? bytes1 : bytes0);
    const uint32_t ld2 = (uint32_t)(second ? a.lda1 : a.lda0) * 2u;
    const uint32_t so = (uint32_t)(cbase - (second ? a.C0 : 0)) * 2u;
    const uint32_t vo = (ppix[i] >= 0 && live) ? (uint32_t)ppix[i] * ld2 + (uint32_t)sc * 16u : PMI_BUF_OOB;
    return buf_load16(rs, vo, so);
  };
  auto load_coef = [&](int chunk) {
    if (PRO) {
      const float* pa = a.pro_a + (int64_t)img * Cin + chunk * CK + sc * 8;
      const float* pb = a.pro_b + (int64_t)img * Cin + chunk * CK + sc * 8;
      *(float4*)ga = *(const float4*)pa; *(float4*)(ga + 4) = *(const float4*)(pa + 4);
      *(float4*)gb = *(const float4*)pb; *(float4*)(gb + 4) = *(const float4*)(pb + 4);
    }
  };
  auto store_piece = [&](char* pbuf, int i, uint4 v) {
    if (PRO) {                                          // GroupNorm-apply + activation; zero padding stays zero
      float f[8];
      unpack8<T>(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = act_apply(f[e] * ga[e] + gb[e], PRO - 1);
      v = pack8<T>(f);
      const uint32_t keep = ppix[i] >= 0 ? 0xffffffffu : 0u;
      v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
    }
    *(uint4*)(pbuf + (tid / CPR) * ROW + sc * 16 + i * (NT / CPR) * ROW) = v;
  };

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  uint4 wq[3][3];                                       // [ring slot][dy]
  auto load_wg = [&](int slot, int group) {             // groups past the end fall outside the resource: zeros (the range check
    const uint32_t vo = wvo + (uint32_t)group * (uint32_t)GB;   // covers the vector offset, so the group offset goes there)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) wq[slot][dy] = buf_load16(rsrc_w, vo + dy * 1024u, 0);
  };

  // ---- prologue: patch of chunk 0, weight groups 0 and 1 ----
  load_wg(0, 0);
  load_wg(1, 1);
  load_coef(0);
  {
    uint4 pr[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) pr[i] = load_piece(0, i, true);
#pragma unroll
    for (int i = 0; i < NPI; ++i) store_piece(smem, i, pr[i]);
  }
  __syncthreads();
  STAMP(1);
  CSTAMP(0);

  const int frag0 = (wm * 8 * PW + l31) * ROW + lhi * 16;   // byte offset of this lane's fragment piece for output row 0, tap dx = 0, k-step 0
  // Rolling fragment registers: output row i of a group needs patch rows i, i+1, i+2, so fragment i is dead once row i's three
  // MFMAs have issued and is reloaded right there with the NEXT group's fragment i -- the reads ride under the MFMAs of the
  // current group (no read phase at the head of a group, no second register set).  The last group of a chunk prefetches from the
  // other patch buffer, so the chunk's only barrier sits in front of that group's first prefetch: by then every wave has issued
  // (and, through the barrier's lgkmcnt(0), received) its last read of the current buffer's successor-to-be-overwritten, and the
  // next patch (staged in groups 1..NPG of this chunk) is complete.
  constexpr int PPG = (NPI + NG - 3) / (NG - 2);       // staging pieces per group (loads in groups 0.., stores one group later)
  constexpr int NPG = (NPI + PPG - 1) / PPG;           // groups that load pieces
  static_assert(NPG <= NG - 2, "the next patch must be complete before the barrier in the chunk's last group");
  uint4 xf[10];
#pragma unroll
  for (int r = 0; r < 10; ++r) xf[r] = *(const uint4*)(smem + frag0 + r * PW * ROW);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const char* const pb = smem + (chunk & 1) * PATCH_BYTES;
    char* const pn = smem + ((chunk + 1) & 1) * PATCH_BYTES;
    const bool more = chunk + 1 < nchunks;              // last chunk: the staging below runs on zeros into the unused buffer (no branches
    const int cn = more ? chunk + 1 : chunk;            // around loads: the compiler then keeps exact vmcnt counts)
    const int gbase = chunk * NG;
    uint4 pr[PPG];
#pragma unroll
    for (int j = 0; j < PPG; ++j) pr[j] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      load_wg((g + 2) % 3, gbase + g + 2);
      if (g == 0) load_coef(cn);
      uint4 prn[PPG];
#pragma unroll
      for (int j = 0; j < PPG; ++j) {
        prn[j] = pr[j];
        if (g < NPG && g * PPG + j < NPI) prn[j] = load_piece(cn, g * PPG + j, more);
      }
      if (g == NG - 1) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);   // keep the global loads in front of the MFMAs (the scheduler sinks them to their use)
      // the GroupNorm-apply + activation of the pieces loaded one group ago (~100 VALU instructions each) is interleaved with this
      // group's 24 MFMAs: an MFMA holds the vector issue port for 8 of its 32 cycles, the rest is free for VALU work
      bool staging = false;
      if (g >= 1 && g <= NPG) {
#pragma unroll
        for (int j = 0; j < PPG; ++j)
          if ((g - 1) * PPG + j < NPI) { store_piece(pn, (g - 1) * PPG + j, pr[j]); staging = true; }
      }
      // next group's fragments: (dx, k-step) of group g+1 in this chunk, or group 0 of the next chunk from the other buffer
      const char* const nb = (g + 1 < NG ? pb : pn) + frag0 + (((g + 1) % NG) / KS) * ROW + (((g + 1) % NG) % KS) * 32;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) acc[i] = T::mfma32(wq[g % 3][dy], xf[i + dy], acc[i]);
        xf[i] = *(const uint4*)(nb + i * PW * ROW);
      }
      xf[8] = *(const uint4*)(nb + 8 * PW * ROW);
      xf[9] = *(const uint4*)(nb + 9 * PW * ROW);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);        // 3 MFMAs of output row i
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // reload fragment i
        if (PRO && staging) __builtin_amdgcn_sched_group_barrier(0x002, WD_ILV * 3 * PPG, 0);   // staging VALU
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int j = 0; j < PPG; ++j) pr[j] = prn[j];
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  CSTAMP(1);
  STAMP(2);
  // ---- epilogue: the whole tile goes through LDS once and leaves as full pixel rows (BN x 2 B contiguous) ----
  char* const stg = smem;
  float* const stat = (float*)(smem + NPX * SROW);     // [NW][BN][2] (sum, sumsq) partials per write-out wave, summed in a fixed order
  float* const bsm = stat + NW * BN * 2;               // [BN] bias + per-sample bias
  for (int c = tid; c < BN; c += NT) {
    const int n = n0 + c;
    float b = 0.f;
    if (a.bias) b = a.bias[n];
    if (a.nbias) b += a.nbias[(int64_t)img * (a.ldnb ? a.ldnb : a.N) + n];
    bsm[c] = b;
  }
  constexpr int LPP = BN / 8;                          // lanes per pixel row at write-out (16 B = 8 channels each)
  constexpr int PPI = 64 / LPP;                        // pixels per store instruction
  constexpr int PXW = NPX / NW;                        // pixels written out per wave
  constexpr int NWI = PXW / PPI;                       // store instructions per wave
  const int q = lane % LPP, psub = lane / LPP;
  const int cl0 = q * 8;
  uint4 rres[NWI];
#pragma unroll
  for (int t = 0; t < NWI; ++t) {                      // residual loads fly while the accumulators are staged
    const int p = wid * PXW + t * PPI + psub;
    const int y = y0 + (p >> 5), x = x0 + (p & 31);
    rres[t] = make_uint4(0, 0, 0, 0);
    if (a.R) {
      const int64_t rr = a.res_up ? (((int64_t)img * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1)) * a.ldr
                                  : (((int64_t)img * a.H + y) * a.W + x) * a.ldr;
      rres[t] = *(const uint4*)((const u16*)a.R + rr + n0 + cl0);
    }
  }
  __syncthreads();                                     // bsm visible (the main loop's last barrier already freed the patch buffers)
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cl = wn * 32 + 8 * g + 4 * lhi;
      const float4 b = *(const float4*)(bsm + cl);
      float v[4] = {acc[i][4 * g] * a.alpha + b.x, acc[i][4 * g + 1] * a.alpha + b.y,
                    acc[i][4 * g + 2] * a.alpha + b.z, acc[i][4 * g + 3] * a.alpha + b.w};
      if (a.act != PMI_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], a.act);
      }
      *(uint2*)(stg + ((wm * 8 + i) * 32 + l31) * SROW + cl * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
  __syncthreads();
  STAMP(6);
  float cs[16];                                        // [0..7] sums, [8..15] sums of squares of this lane's 8 channels
#pragma unroll
  for (int e = 0; e < 16; ++e) cs[e] = 0.f;
#pragma unroll
  for (int t = 0; t < NWI; ++t) {
    const int p = wid * PXW + t * PPI + psub;
    uint4 v = *(const uint4*)(stg + p * SROW + cl0 * 2);
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
    const int y = y0 + (p >> 5), x = x0 + (p & 31);
    *(uint4*)((u16*)a.D + (((int64_t)img * a.H + y) * a.W + x) * a.ldd + n0 + cl0) = v;
  }
  STAMP(7);
  if (a.stats) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (LPP <= 16) cs[e] += __shfl_xor(cs[e], 16);
      cs[e] += __shfl_xor(cs[e], 32);
    }
    if (lane < LPP) {
      float* const slot = stat + (wid * BN + cl0) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) { slot[2 * e] = cs[e]; slot[2 * e + 1] = cs[8 + e]; }
    }
    __syncthreads();
    float* o = a.stats + (((int64_t)img * a.stats_p + ty * tiles_x + tx) * a.N + n0) * 2;
    for (int c = tid; c < 2 * BN; c += NT) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += stat[w * 2 * BN + c];
      o[c] = v;
    }
  }
#ifdef PMI_STAMPS
  __syncthreads();
  STAMP(3);
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
  if (cfg == 4) {          // 8x32 px x 256 ch, 8 waves, one workgroup per CU
    const int tiles = nimg * (a.H / 8) * (a.W / 32) * (a.N / 256);
    hipLaunchKernelGGL((conv3x3_wd_kernel<T, 8, 1, 64, PRO>), dim3(tiles), dim3(512), 0, s, a);
  } else {                 // 8x32 px x 128 ch, 4 waves, two workgroups per CU
    const int tiles = nimg * (a.H / 8) * (a.W / 32) * (a.N / 128);
    hipLaunchKernelGGL((conv3x3_wd_kernel<T, 4, 1, 32, PRO>), dim3(tiles), dim3(256), 0, s, a);
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

int pmi_conv3x3_wd_launch(const pmi_igemm_args* a, int cfg, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  return a->dtype == PMI_DT_BF16 ? launch_t<BF16>(*a, s, cfg) : launch_t<F16>(*a, s, cfg);
}
