// conv3x3 (stride 1, pad 1) for gfx950 MFMA, "weights-direct" form -- the dominant kernel of the UNets.
//
// A 512-thread workgroup owns 8 image rows x 32 pixels x 256 output channels; every wave owns a 256-pixel x 32-channel
// tile (128 accumulator registers) and streams ITS OWN weights global -> VGPR, already in MFMA fragment order (packed once on the
// host: engine/ops.py PackedLinear.frag / frag16): one coalesced 1 KB buffer load per fragment, no LDS and no cross-wave
// synchronisation on the weight side.  Only the activation patch goes through LDS: per 64-channel input chunk the 10 x 34 halo
// patch is staged once (double-buffered) and serves all 9 taps of all 8 waves -> ONE workgroup barrier per chunk (288
// 32x32x16 MFMAs per wave between barriers; the round-1 kernel had one barrier per tap).
//
// Loop order inside a chunk is (dx, channel step) with the three dy taps innermost: the 8 output rows of a wave need the same 10
// patch-row fragments for dy = 0, 1, 2, so a group reads 10 fragments from LDS for 24 (32x32x16) or 48 (16x16x32) MFMAs
// (0.42 ds_read_b128 per 32-cycle MFMA slot; round 1: 0.75).  The fragments live in a rolling window of SIX registers (see below),
// weights in a small register ring fed one or two groups ahead.
//
// MF16 selects v_mfma_f32_16x16x32 instead of 32x32x16: same FLOPs per cycle, but the chip holds a higher clock on it (this
// kernel is power-limited: 1.4-1.65 GHz in the main loop).  The wave's tile is then 8 rows x 2 half-rows of 16 pixels x 2 blocks of
// 16 channels, a k-step covers 32 channels, a group is (dx, 32-channel step, half-row) and the 6 weight fragments of a (dx, step)
// pair serve both half-rows.
//
// LDS patch layout: one row per patch pixel, 64 channels + padding, NOT swizzled: a fragment read is (per-lane base) +
// (compile-time offset).  Pitch 144 B (32x32x16: the 16 lanes of a ds_read_b128 group start at dwords 36 i) or 160 B (16x16x32: the
// 16-byte slots 10 i + k-quarter): 16 different slots for 16 consecutive pixels at every tap shift -> conflict-free.
//
// Fused: nearest-x2 upsample of the input (patch gather), skip-concat (two sources), GroupNorm-apply(+FiLM)+activation on
// the patch as it is written to LDS (zero padding stays zero); bias / per-sample bias / activation / residual (optionally
// through a nearest-x2 upsample) / per-channel (sum, sumsq) of the output for the next GroupNorm in the epilogue, which
// transposes the whole tile through LDS and writes full pixel rows (512 contiguous bytes).
#include "common.h"
#include "../../include/perceptor_hip.h"

#ifndef WLG
#define WLG 1        // which group of a (dx, step) pair issues the next pair's weight loads: 0 = the first (two groups of latency cover), 1 = the second
#endif
#ifndef WD_ILV_SIN
#define WD_ILV_SIN 4 // the same for the split-input staging (two loads, ~115 VALU instructions per unit): same-box A/B on the mixed c5 step: 2: 61.7 ms, 3: 60.9, 4: 60.7 (bf16 with 4: +0.7 ms, hence its own value)
#endif
#ifndef WD_FINE
#define WD_FINE 2
#endif
#ifndef WD_DIAG
#define WD_DIAG 0    // diagnostic builds only (tools/conv_decomp.sh; results are wrong, timing is the point): 1 no weight loads in the main loop, 2 no fragment reads, 4 no patch staging, 8 no chunk barrier, 16 patch loads out of range (no memory latency, conversion kept)
#endif
#ifndef WD_RD
#define WD_RD 4   // split epilogue: residual prefetch depth. 6 / 8 put 72-192 B per lane of scratch into these kernels (some of it in the main loop)
#endif
#ifndef WD_EARLY
#define WD_EARLY 0   // 1: the next piece's load issued in the store slot itself (two groups of cover instead of one, no extra registers). Measured, same box: 256 -> 256 @256x256 0.488 -> 0.511 ms, bf16 c5 39.9 -> 40.5 ms, mixed 55.5 -> 57.1 -- slower: the wait for the carried piece then opens the group, in front of every MFMA
#endif
#ifndef WD_DEFER_SIN
#define WD_DEFER_SIN 1
#endif
#ifndef WD_FINE_SIN
#define WD_FINE_SIN 2   // mixed c5 step, same box: 3: 60.8-61.0 ms, 2: 59.6, row-level hints (0): 60.3, write in front of the rows: 61.7
#endif
#ifndef WD_ILV
#define WD_ILV 2     // (4 until the last sweep: 2 leaves the 128-channel tiles 12 instead of 32 B/lane of scratch and measures +2 % there) VALU instructions of the patch staging the scheduler is asked to place behind each output row's MFMAs
#endif

namespace {

constexpr int PW = 34;
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// NWN waves x 32 channels = the tile's output channels; CK = input channels per staged chunk.  (8, 64): one 8-wave workgroup per CU,
// 256-channel tiles (configs 4 / 6).  (4, 32): 128-channel tiles for the 128-channel layers, 4 waves and ~75 KB of LDS, so TWO
// workgroups share a CU and one's prologue / epilogue runs under the other's main loop (config 7; 16x16x32 MFMA only).
// SPL: precise mode (dtype 2): the output (and a residual) are hi + lo f16 pairs, [C/32][hi 32 | lo 32] per pixel (common.h: F16X2); the
// input needs nothing special -- its 2C physical channels are an ordinary K dimension against duplicated weights.
// SK: split-K over grid.z (its own instantiation: the chunk-range variables cost the unsplit kernel registers it does not have)
// SMALLC (config 8): at most 32 input channels (the UNets' first convolution: 3 image channels padded to 8, SD's 4 latent channels, yfcc's 19 -> 24):
// the whole K = 9 taps x Cin fits a handful of 32-deep MFMA steps, so the patch (10 x 34 pixels x Cin) is staged once, the wave's weights sit in
// registers, and the B fragments are gathered per tap straight from the patch; tile, accumulator layout and epilogue are config 7's.
// SIN (mixed mode, split_in): the sources are hi + lo tensors ([C/32][hi 32 | lo 32] per pixel) AND a prologue is fused, which needs the
// VALUE hi + lo of a channel: a staging unit is then the pair of 16-byte pieces holding the high and the low parts of 8 logical channels.
//   SIN = 2: single operand: act((hi + lo) * a + b) is rounded ONCE to f16 -- the K loop is the plain kernel's over the logical channels.
//   SIN = 1: doubled operand: the result is split again, y = yh + yl, and a chunk's K is [yh of CK/2 logical channels | their yl] against
//            weights duplicated in the same pattern (W*yh + W*yl in fp32: an fp32-grade product at twice the MFMA work).
// (PRO = 0 over a split input needs neither: its 2C physical channels are an ordinary K dimension, SIN = 0.)
template <typename T, int PRO, bool MF16, int NWN = 8, int CK = 64, bool SPL = false, bool SK = false, bool SMALLC = false, int SIN = 0>
__global__ __launch_bounds__(NWN * 64, 2) void conv3x3_wd_kernel(const pmi_igemm_args a) {
  static_assert(SIN == 0 || (PRO != 0 && MF16 && !SK && !SMALLC && std::is_same<T, F16>::value), "split-input staging: fused prologue, f16, 16x16x32 tiles");
  constexpr int NW = NWN, NT = NW * 64;
  constexpr int KS = MF16 ? CK / 32 : CK / 16;         // k-steps (one MFMA deep) per chunk
  constexpr int PP = 10 * PW;                          // patch pixels
  constexpr int CPR = SIN == 1 ? CK / 16 : CK / 8;     // staging units per patch pixel: 16-byte pieces, or (hi, lo) pairs of them
  constexpr int LCK = SIN == 1 ? CK / 2 : CK;          // logical input channels per chunk
  constexpr int NLD = SIN ? 2 : 1;                     // 16-byte loads per staging unit
  constexpr int NG = MF16 ? 3 * KS * 2 : 3 * KS;       // groups per chunk: 3 dx x k-steps (x 2 half-rows with 16x16x32); 12 at CK = 64
  constexpr int WGC = 3 * KS;                          // weight groups per chunk (3 dy fragments of 1 KB each, x 2 channel blocks with MF16)
  constexpr int NWF = MF16 ? 6 : 3;                    // fragments per weight group
  constexpr int GB = NWF * 1024;
  constexpr int NPI = (PP * CPR + NT - 1) / NT;        // 16-byte staging pieces per thread per chunk (6)
  constexpr int ROW = CK * 2 + ((MF16 && CK == 64) ? 32 : 16);   // patch row pitch in bytes: 144 / 160 / 80, conflict-free for the b128 fragment reads
  constexpr int PATCH_BYTES = (NPI * NT / CPR) * ROW;  // padded to whole staging passes: no bounds test on the LDS writes
  constexpr int BN = NWN * 32, NPX = 256;
  constexpr int SROW = BN * 2 + 16;                    // epilogue staging row: BN 16-bit channels + 16 B pad
  constexpr int FROW = (BN / 2) * 4 + 16;              // split epilogue: fp32 image of HALF the tile's channels per pass, 16 B pad
  constexpr int EPI_BYTES = SPL ? NPX * FROW + NW * (BN / 2) * 8 + BN * 4 : NPX * SROW + NW * BN * 8 + BN * 4;
  constexpr int MAXCIN = NWN == 8 ? 2048 : 1024;       // fused-prologue coefficient table: a[Cin], b[Cin] fp32 of this image
  constexpr int CTAB = MAXCIN + 8;                     // one coefficient table: MAXCIN channels + a unit of eight zeros (padding pixels, see read_coef)
  constexpr int COEF_BYTES = PRO ? 2 * CTAB * 4 : 0;
  constexpr int PPIX_BYTES = NPI * NT * 4;             // source pixel of every staged piece of this thread (kept out of the registers)
  // Staging of the next patch: store slot k (k = 0..NPI-1) is group SG k; the piece it stores was loaded one slot earlier into
  // the single carried register set (piece 0 in the previous chunk's last group).  With MF16 the store slots are the half-row-0
  // groups, where only ONE weight set is live, so the prologue's temporaries fit the register file.
  constexpr int SG = (MF16 && 2 * (NPI - 1) <= NG - 2) ? 2 : 1;
  static_assert(SG * (NPI - 1) <= NG - 1, "the next patch must be complete before the barrier in the chunk's last group");
  // bias + per-sample bias of the tile's BN channels: filled in the prologue (its global latency hides under the first patch loads), read by
  // the epilogue -- which then starts without a load and without a barrier of its own; behind both the main loop's and the epilogue's regions
  constexpr int BSM_OFF = cmax(2 * PATCH_BYTES + COEF_BYTES + PPIX_BYTES, EPI_BYTES - BN * 4);
  static_assert(BSM_OFF + BN * 4 <= (NWN == 8 ? 160 : 80) * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem[BSM_OFF + BN * 4];
  float* const bsm = (float*)(smem + BSM_OFF);
  float* const coef = (float*)(smem + 2 * PATCH_BYTES);
  int* const ppix_s = (int*)(smem + 2 * PATCH_BYTES + COEF_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  if (PRO && tid < 16) coef[(tid >> 3) * CTAB + MAXCIN + (tid & 7)] = 0.f;     // the zero unit of both tables (visible after the prologue's barrier)
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wn = wid;

  const int tiles_x = a.W / 32, tiles_y = a.H / 8, tiles_n = (a.N + BN - 1) / BN;   // Cout a multiple of 32: the last tile's waves past Cout multiply zeros and store nothing
  const int nimg = a.M / (a.H * a.W);
  int logical = xcd_remap(blockIdx.x, nimg * tiles_y * tiles_x * tiles_n);
  const int tn = logical % tiles_n; logical /= tiles_n;
  const int tx = logical % tiles_x; logical /= tiles_x;
  const int ty = logical % tiles_y;
  const int img = logical / tiles_y;
  const int y0 = ty * 8, x0 = tx * 32, n0 = tn * BN;

#ifdef PMI_STAMPS   // tools/conv_probe.py --stamps: phase timestamps (100 MHz) per workgroup, shader clock per wave around the main loop
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
  // split-K (grid.z): this workgroup reduces chunks [cb0, ce) and leaves raw fp32 sums in its slab of a.ws (maps too small to fill the chip
  // with output tiles alone: 32x32 at batch 8); bias / activation / residual then happen in splitk_reduce_kernel (igemm.hip)
  const int nsplit = SK ? a.splitk : 1;
  const int cb0 = SK ? (nchunks / nsplit) * blockIdx.z : 0, ce = SK ? cb0 + nchunks / nsplit : nchunks, nloc = ce - cb0;
  const int sc = tid % CPR;                            // staging unit (8 channels) of a patch pixel this thread stages
  const int64_t img_px = (int64_t)a.Hin * a.Win;
  const u16* const A0i = (const u16*)a.A0 + (int64_t)img * img_px * a.lda0;
  const u16* const A1i = a.A1 ? (const u16*)a.A1 + (int64_t)img * img_px * a.lda1 : A0i;
  // a.C0 / a.C1 count what the K index counts: logical channels for SIN = 2 (rows hold twice as many values), physical ones otherwise
  const int64_t bytes0 = ((img_px - 1) * a.lda0 + (SIN == 2 ? 2 * a.C0 : a.C0)) * 2;
  const int64_t bytes1 = a.A1 ? ((img_px - 1) * a.lda1 + (SIN == 2 ? 2 * a.C1 : a.C1)) * 2 : 0;
  const int C0l = SIN == 1 ? a.C0 >> 1 : a.C0;          // logical channels of the first source
  const int CinL = SIN == 1 ? Cin >> 1 : Cin;           // logical input channels (the prologue's coefficient count)
  // this wave's weight stream: [chunk][dx][step][dy](x [16-channel block])[lane][8] 16-bit, contiguous in loop order
  const int64_t wslab = (int64_t)nchunks * WGC * GB;
  const bool wave_live = n0 + wn * 32 < a.N;           // wave-uniform; a dead wave's weight resource is empty (every load returns zeros)
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc((const char*)a.Bf + (wave_live ? (int64_t)(tn * NWN + wn) * wslab : 0), wave_live ? wslab : 0);
  const uint32_t wvo = (uint32_t)lane * 16u;

  // ---- staging plan: source pixel (inside the image) per staged piece, -1 = zero padding; lives in LDS ([piece][thread]:
  // conflict-free dword reads), as do the prologue's per-channel coefficients -- the main loop has no registers to spare ----
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int pp = tid / CPR + (NT / CPR) * i;
    const int py = pp / PW, px = pp - py * PW;
    const int sy = y0 - 1 + py, sx = x0 - 1 + px;
    const bool inside = pp < PP && (unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W;
    ppix_s[i * NT + tid] = inside ? (sy >> a.up) * a.Win + (sx >> a.up) : -1;
  }
  // (the coefficient table is filled in the prologue below, AFTER the first patch loads are in flight: one memory latency, not two)

  // source of chunk `chunk`: which tensor, row pitch, channel offset (wave-uniform: C0 is a multiple of CK)
  struct Piece { uint4 v[NLD]; };
  auto load_piece = [&](int chunk, int pix, bool live) -> Piece {       // !live: out-of-range offset, zeros, no traffic, no branch
    Piece r;
    if constexpr (SIN == 0) {
      const int cbase = chunk * CK;
      const bool second = cbase >= a.C0;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(second ? A1i : A0i, second ? bytes1 : bytes0);
      const uint32_t ld2 = (uint32_t)(second ? a.lda1 : a.lda0) * 2u;
      const uint32_t so = (uint32_t)(cbase - (second ? a.C0 : 0)) * 2u;
      const uint32_t vo = (pix >= 0 && live) ? (uint32_t)pix * ld2 + (uint32_t)sc * 16u : PMI_BUF_OOB;
      r.v[0] = buf_load16(rs, vo, so);
    } else {
      // logical channels [cl, cl + LCK) of one source (cl a multiple of LCK = 16, 32 or 64: a unit's 8 channels never straddle a 32-group
      // unless LCK = 64, where units 4..7 sit in the next group -- split_off of the per-thread part alone covers both cases)
      const int cbase = chunk * LCK;
      const bool second = cbase >= C0l;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(second ? A1i : A0i, second ? bytes1 : bytes0);
      const uint32_t ld2 = (uint32_t)(second ? a.lda1 : a.lda0) * 2u;
      const int cl = cbase - (second ? C0l : 0);
      const uint32_t so = (uint32_t)split_off(cl, 32) * 2u;
      const uint32_t vo = (pix >= 0 && live) ? (uint32_t)pix * ld2 + (uint32_t)split_off(sc * 8, 32) * 2u : PMI_BUF_OOB;
      r.v[0] = buf_load16(rs, vo, so);
      r.v[1] = buf_load16(rs, vo + 64u, so);            // the low parts: 32 elements behind the high parts (an out-of-range offset stays out of range)
    }
    return r;
  };
  // Coefficients of one staging unit.  A padding pixel (pix < 0) reads zeros instead: its loaded values are zeros too, so act(0 * 0 + 0) = 0
  // is the zero padding itself -- one address select per unit instead of masking the packed result (4-8 v_and beside the MFMAs).
  // For SiLU the table holds a' = -log2(e) a, b' = -log2(e) b: u = a' x + b' = -log2(e) t feeds v_exp directly and
  // silu(t) = t / (1 + e^-t) = u / (-(1 + 2^u) / ln 2) = u * rcp(fma(2^u, k, k)), k = -log2(e): fma, exp, fma, rcp, mul -- five vector
  // instructions per value instead of six.  (The 128-channel tiles are vector-ISSUE bound: per 16x16x32 MFMA the SIMD has 8 spare issue
  // cycles, MI355X_MICROARCH.md per-instruction constants, and their staging used ~105 % of them.)
  constexpr bool PSILU = PRO == 1 + PMI_ACT_SILU;
  constexpr float NLOG2E = -1.4426950408889634f;
  // (split-input instantiations keep the mask: the select's extra register tipped them into in-loop scratch reloads, each a full vmcnt drain)
  constexpr bool ZPAD = SIN == 0;
  constexpr bool DEFER = PRO != 0 && (SIN == 0 || WD_DEFER_SIN);
  auto read_coef = [&](int chunk, int pix, float* ga, float* gb) {
    const float* ca = coef + ((!ZPAD || pix >= 0) ? chunk * LCK + sc * 8 : MAXCIN);
    *(float4*)ga = *(const float4*)ca; *(float4*)(ga + 4) = *(const float4*)(ca + 4);
    *(float4*)gb = *(const float4*)(ca + CTAB); *(float4*)(gb + 4) = *(const float4*)(ca + CTAB + 4);
  };
  // GroupNorm-apply + activation of one staged unit (registers only), and its LDS write.  The main loop issues the two apart: the compiler
  // cannot tell the patch buffers from each other, so every fragment read behind a write in program order waits for it -- and with the
  // write in front of a group's MFMA rows the whole conversion had to retire before the group's second fragment read, in ONE gap between
  // two MFMAs (~75-115 vector instructions; the interleave hints below had nothing left to place).  The write goes behind the rows.
  auto unpack_piece = [&](const Piece& pc, float* f) {   // the unit's 8 values; the piece's registers are dead behind this
    unpack8<T>(pc.v[0], f);
    if constexpr (SIN != 0) {
      float lo[8];
      unpack8<T>(pc.v[1], lo);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += lo[e];
    }
  };
  auto finish_piece = [&](float* f, int pix, const float* ga, const float* gb, bool mask) -> Piece {
    Piece o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (PSILU) {
        const float u = __builtin_fmaf(f[e], ga[e], gb[e]);
        f[e] = u * __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(u), NLOG2E, NLOG2E));
      } else {
        f[e] = act_apply(f[e] * ga[e] + gb[e], PRO - 1);
      }
    }
    // mask: the prologue's first patch shares ONE coefficient read between its units, so its padding units are masked here; the main
    // loop's units read zero coefficients instead (read_coef)
    const uint32_t keep = (!mask || pix >= 0) ? 0xffffffffu : 0u;
    uint4 v = pack8<T>(f);
    if constexpr (SIN == 1) {                         // second operand half: what the 16-bit value lost
      float lo[8];
      float hf[8];
      unpack8<T>(v, hf);
#pragma unroll
      for (int e = 0; e < 8; ++e) lo[e] = f[e] - hf[e];
      uint4 vl = pack8<T>(lo);
      vl.x &= keep; vl.y &= keep; vl.z &= keep; vl.w &= keep;
      o.v[1] = vl;
    } else if constexpr (SIN == 2) {
      o.v[1] = v;
    }
    v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
    o.v[0] = v;
    return o;
  };
  auto cvt_piece = [&](Piece pc, int pix, const float* ga, const float* gb, bool mask) -> Piece {
    if (!PRO) return pc;
    float f[8];
    unpack_piece(pc, f);
    return finish_piece(f, pix, ga, gb, mask);
  };
  auto write_piece = [&](char* pbuf, int i, const Piece& o) {
    char* const dst = pbuf + (tid / CPR) * ROW + sc * 16 + i * (NT / CPR) * ROW;
    *(uint4*)dst = o.v[0];
    if constexpr (PRO != 0 && SIN == 1) *(uint4*)(dst + CK) = o.v[1];   // [yh of the chunk's CK / 2 channels | their yl]: CK bytes apart
  };

  f32x16 acc[MF16 ? 1 : 8];                             // 32x32x16: [row]
  f32x4 acc4[MF16 ? 8 : 1][2][2];                       // 16x16x32: [row][half-row][16-channel block]
#pragma unroll
  for (int i = 0; i < (MF16 ? 1 : 8); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (MF16 ? 8 : 1); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc4[i][(r >> 3) & 1][(r >> 2) & 1][r & 3] = 0.f;

  if (!SK) {
    for (int c = tid; c < BN; c += NT) {                 // (made visible by the prologue's barrier)
      const int n = n0 + c;
      float b = 0.f;
      if (n < a.N) {
        if (a.bias) b = a.bias[n];
        if (a.nbias) b += a.nbias[(int64_t)img * (a.ldnb ? a.ldnb : a.N) + n];
      }
      bsm[c] = b;
    }
  }
  if constexpr (SMALLC) {
    static_assert(MF16 && NWN == 4 && PRO == 0 && !SK, "config 8 is a 128-channel-tile instantiation");
    const int Cp = a.C0, C8 = Cp >> 3;                     // 8 .. 32 input channels, one source
    const int KSS = (9 * Cp + 31) >> 5;                    // 32-deep MFMA steps over k = tap * Cp + c (3 .. 9); weights are zero past 9 * Cp
    const u16* const Ai = (const u16*)a.A0 + (int64_t)img * a.Hin * a.Win * a.lda0;
    for (int p = tid; p < PP * C8; p += NT) {              // the patch, [pixel][Cp] 16-bit, zero outside the image
      const int pp = p / C8, c8 = p - pp * C8;
      const int py = pp / PW, px = pp - py * PW;
      const int sy = y0 - 1 + py, sx = x0 - 1 + px;
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W) v = *(const uint4*)(Ai + ((int64_t)sy * a.W + sx) * a.lda0 + c8 * 8);
      *(uint4*)(smem + (pp * Cp + c8 * 8) * 2) = v;
    }
    // this wave's weights, host-packed (PackedLinear.frag_c8): [N/32][KSS][16-channel block (2)][lane][8]
    uint4 wsm[9][2];
    int boff[9];                                           // byte offset of the lane's 8 input channels of step s inside the patch, relative to its pixel
    const bool wave_ok = n0 + wn * 32 < a.N;
    const char* const wb = (const char*)a.Bf + (int64_t)(tn * NWN + wn) * KSS * 2048 + lane * 16;
#pragma unroll
    for (int s_ = 0; s_ < 9; ++s_) {
      wsm[s_][0] = wsm[s_][1] = make_uint4(0, 0, 0, 0);
      boff[s_] = 0;
      if (s_ < KSS) {
        if (wave_ok) { wsm[s_][0] = *(const uint4*)(wb + s_ * 2048); wsm[s_][1] = *(const uint4*)(wb + s_ * 2048 + 1024); }
        const int k0 = 32 * s_ + 8 * (lane >> 4);
        int tap = k0 / Cp;
        const int c0 = k0 - tap * Cp;
        tap = tap > 8 ? 8 : tap;                           // past the last tap the weights are zero: any valid address
        boff[s_] = (((tap / 3) * PW + (tap % 3)) * Cp + c0) * 2;
      }
    }
    __syncthreads();
    const int pix0 = (lane & 15) * Cp * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        const char* const pb_ = smem + pix0 + ((i * PW + sx * 16) * Cp) * 2;
#pragma unroll
        for (int s_ = 0; s_ < 9; ++s_) {
          if (s_ < KSS) {
            const uint4 bf = *(const uint4*)(pb_ + boff[s_]);
            acc4[i][sx][0] = T::mfma16(wsm[s_][0], bf, acc4[i][sx][0]);
            acc4[i][sx][1] = T::mfma16(wsm[s_][1], bf, acc4[i][sx][1]);
          }
        }
      }
    __syncthreads();                                       // every wave is done with the patch before the epilogue reuses the LDS
  } else {
  uint4 wq[MF16 ? 2 : 3][NWF];                          // [ring slot][dy (x channel block)]
  auto load_wg = [&](int slot, int group) {             // groups past the end fall outside the resource: zeros (the range check
    const uint32_t vo = wvo + (uint32_t)group * (uint32_t)GB;   // covers the vector offset, so the group offset goes there)
#pragma unroll
    for (int f = 0; f < NWF; ++f) wq[slot][f] = buf_load16(rsrc_w, vo + f * 1024u, 0);
  };

  // ---- prologue: patch of chunk 0, the first weight groups ----
  load_wg(0, cb0 * WGC);
  if (!MF16 || (WD_DIAG & 1)) load_wg(1, cb0 * WGC + 1);
  {
    Piece p0[NPI];
    float ga[8], gb[8];
#pragma unroll
    for (int i = 0; i < NPI; ++i) p0[i] = load_piece(cb0, ppix_s[i * NT + tid], true);    // (a thread reads only its own table entries)
    if (PRO) {
      for (int c = tid; c < CinL; c += NT) {
        coef[c] = a.pro_a[(int64_t)img * CinL + c] * (PSILU ? NLOG2E : 1.f);
        coef[CTAB + c] = a.pro_b[(int64_t)img * CinL + c] * (PSILU ? NLOG2E : 1.f);
      }
      __syncthreads();
      read_coef(cb0, 0, ga, gb);
    }
#pragma unroll
    for (int i = 0; i < NPI; ++i) write_piece(smem, i, cvt_piece(p0[i], ppix_s[i * NT + tid], ga, gb, true));
  }
  __syncthreads();
  STAMP(1);
  CSTAMP(0);

  // byte offset of this lane's fragment piece for output row 0, tap dx = 0, k-step 0 (half-row 0)
  const int frag0 = MF16 ? (lane & 15) * ROW + (lane >> 4) * 16 : l31 * ROW + lhi * 16;
  // byte offset of group g's fragments relative to frag0: 32x32x16: (dx, 16-channel step); 16x16x32: (dx, 32-channel step, half-row)
  auto goff = [&](int g) -> int {
    if (MF16) { const int q = g >> 1, s = g & 1; return (q / KS) * ROW + (q % KS) * 64 + s * 16 * ROW; }
    return (g / KS) * ROW + (g % KS) * 32;
  };
  // Six fragment registers as a rolling window over the 10 patch rows of a group: row i's MFMAs need rows i, i+1, i+2; once they
  // have issued, row i's register is reloaded -- with row i+6 of this group (i < 4), or with row i-4 of the NEXT group (i = 4, 5, 6;
  // rows 3, 4, 5 of the next group replace rows 7, 8, 9 after the last output row).  Row r of group g therefore sits in register
  // (r + 4 g) % 6; 4 NG is a multiple of 6, so the assignment repeats per chunk and every index is a compile-time constant.
  // The last group of a chunk reads its own rows 6..9 from the current buffer (after output rows 0..3) and then prefetches the next
  // chunk's rows from the other buffer: the chunk's only barrier sits between the two.
  static_assert((4 * NG) % 6 == 0, "fragment window must repeat per chunk");
  uint4 xf[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) xf[r] = *(const uint4*)(smem + frag0 + r * PW * ROW);
  // carried staging state: the piece waiting for its store slot, its source pixel, and the source pixel of the next load.
  // Table reads (ppix_s, coef) are issued a phase before their use, so their waits are counted lgkmcnt(N), not drains.
  int pixc = ppix_s[0 * NT + tid];
  Piece pr = load_piece(nloc > 1 ? cb0 + 1 : cb0, pixc, nloc > 1);    // piece 0 of the second chunk's patch
  int pixn = ppix_s[1 * NT + tid];
  // With an odd number of weight groups per chunk (CK = 32: 3) the two-slot weight ring changes phase from chunk to chunk; the ring slot
  // must be a compile-time register index, so the chunk body is unrolled over both phases and the loop advances two chunks at a time
  // (Cin is a multiple of 64: the chunk count is even).
  constexpr int CSTEP = (MF16 && (WGC & 1)) ? 2 : 1;
  for (int chunk0 = cb0; chunk0 < ce; chunk0 += CSTEP) {
#pragma unroll
   for (int ph = 0; ph < CSTEP; ++ph) {                  // fully unrolled: PH is a constant in each copy of the body
    const int chunk = chunk0 + ph;
    const int PH = (ph * WGC) & 1;                       // parity of this chunk's first weight-group index
    const char* const pb = smem + ((chunk - cb0) & 1) * PATCH_BYTES;
    char* const pn = smem + ((chunk - cb0 + 1) & 1) * PATCH_BYTES;
    const bool more = chunk + 1 < ce;                   // past the end the staging runs on zeros into the unused buffer (no branches
    const int cn = more ? chunk + 1 : chunk;            // around loads: the compiler then keeps exact vmcnt counts)
    const bool more2 = chunk + 2 < ce;
    const int cn2 = more2 ? chunk + 2 : chunk;
    const int gbase = chunk * WGC;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (WD_DIAG & 1) {}
      else if (MF16) { if ((g & 1) == WLG) load_wg(((g >> 1) + 1 + PH) & 1, gbase + (g >> 1) + 1); }     // the next (dx, step) pair: issued in the pair's second group (WLG 1), so that the store-slot groups hold ONE weight set
      else load_wg((g + 2) % 3, gbase + g + 2);                                              // two groups ahead
      const bool store_slot = !(WD_DIAG & 4) && g % SG == 0 && g / SG < NPI;            // stores piece g / SG from `pr`
      // EARLY (off, see WD_EARLY): the next piece's load issued in the store slot itself, as soon as the carried piece is unpacked.
      // (Tried because WD_DIAG 16 -- patch loads that return zeros at once -- runs 10-17 % faster; but that build multiplies zeros, draws
      // less power and clocks higher: not a latency measurement.  profiles/r03_conv_decomposition.txt.)
      constexpr bool EARLY = WD_EARLY && SG == 2;
      const int lp = EARLY ? (g / SG + 1) % NPI : (g + 1 == NG) ? 0 : (g - (SG - 1)) / SG + 1;     // piece whose load is issued in this group (if load_slot)
      const bool load_slot = !(WD_DIAG & 4) && (EARLY ? store_slot : ((g + 1 == NG) || (g % SG == SG - 1 && lp < NPI)));
      float ga[8], gb[8], fu[8];
      if (PRO && store_slot) read_coef(cn, pixc, ga, gb);
      if (EARLY && store_slot) {
        if (PRO) unpack_piece(pr, fu);
        else write_piece(pn, g / SG, pr);               // no prologue: the piece goes out as it is
      }
      Piece prn = pr;
      int pixl = pixc;
      if (load_slot) { prn = load_piece(lp == 0 ? cn2 : cn, pixn, !(WD_DIAG & 16) && (lp == 0 ? more2 : more)); pixl = pixn; }   // (piece 0 belongs to the chunk after next; diagnostic 16: the loads return zeros at once)
      __builtin_amdgcn_sched_barrier(0);   // keep the global loads in front of the MFMAs (the scheduler sinks them to their use)
      // GroupNorm-apply + activation of the piece loaded a slot ago, interleaved with this group's MFMAs by the hints below
      Piece po;
      if (store_slot) {
        if (PRO) po = EARLY ? finish_piece(fu, pixc, ga, gb, !ZPAD) : cvt_piece(pr, pixc, ga, gb, !ZPAD);
        if (!DEFER && !(EARLY && !PRO)) write_piece(pn, g / SG, PRO ? po : pr);   // (no prologue: nothing to interleave, the registers go back at once)
      }
      const char* const cb0 = pb + frag0 + goff(g);                                // this group's fragments
      const char* const nb = (g + 1 < NG ? pb : pn) + frag0 + goff((g + 1) % NG);  // next group's (next chunk: the other buffer)
      auto rows = [&](int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i) {
          if constexpr (MF16) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb)
                acc4[i][g & 1][cb] = T::mfma16(wq[((g >> 1) + PH) & 1][dy * 2 + cb], xf[(i + dy + 4 * g) % 6], acc4[i][g & 1][cb]);
          } else {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) acc[i] = T::mfma32(wq[g % 3][dy], xf[(i + dy + 4 * g) % 6], acc[i]);
          }
          if (WD_DIAG & 2) {}
          else if (i < 4) xf[(i + 4 * g) % 6] = *(const uint4*)(cb0 + (i + 6) * PW * ROW);
          else if (i < 7) xf[(i + 4 * g) % 6] = *(const uint4*)(nb + (i - 4) * PW * ROW);
          else {
#pragma unroll
            for (int r = 3; r < 6; ++r) xf[(r + 4 + 4 * g) % 6] = *(const uint4*)(nb + r * PW * ROW);
          }
        }
#pragma unroll
        for (int i = i0; i < i1; ++i) {
          if (WD_FINE && MF16 && PRO && store_slot && i >= (g == NG - 1 ? 1 : 2)) {
            // staging VALU two (split input: three) at a time behind each MFMA: a 16x16x32 MFMA leaves the SIMD 8 issue cycles
#pragma unroll
            for (int m = 0; m < 6; ++m) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, SIN ? WD_FINE_SIN : WD_FINE, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          } else {
          __builtin_amdgcn_sched_group_barrier(0x008, MF16 ? 6 : 3, 0);   // the MFMAs of output row i
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);              // reload the freed fragment register
          if (PRO && store_slot && i >= 2) __builtin_amdgcn_sched_group_barrier(0x002, (SIN ? WD_ILV_SIN : WD_ILV) * 4, 0);   // staging VALU, once the coefficients are in
          }
        }
      };
      if (g == NG - 1) {
        // the chunk's barrier: rows 0..3 of this group issue the wave's LAST reads of the current patch buffer (which the next
        // chunk's staging overwrites); rows 4..7 prefetch from the other buffer, whose staging (slots 0..NPI-1) every wave has finished
        rows(0, 4);
        if (DEFER && store_slot) write_piece(pn, g / SG, po);      // (a store slot in the barrier group, 128-channel tiles: the write belongs in front of the barrier)
        __builtin_amdgcn_sched_barrier(0);
        if (!(WD_DIAG & 8)) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        rows(4, 8);
      } else {
        rows(0, 8);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                // rows 8, 9 were freed by the last output row as well
      if (DEFER && store_slot && g != NG - 1) write_piece(pn, g / SG, po);
      // source pixel of the piece the NEXT load slot fetches (read now, used a group or two later)
      int pixn2 = pixn;
      if (load_slot) pixn2 = ppix_s[((lp + 1) % NPI) * NT + tid];
      pr = prn; pixc = pixl; pixn = pixn2;
      __builtin_amdgcn_sched_barrier(0);
    }
   }
  }

  }
  CSTAMP(1);
  STAMP(2);
  if constexpr (MF16 && !SPL && SK) {
    {                        // raw partial sums: a lane holds 4 consecutive channels of a pixel (16 bytes fp32), quarter-waves complete 64-byte runs
      float* const slab = (float*)a.ws + (int64_t)blockIdx.z * a.M * a.N;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int n = n0 + wn * 32 + cb * 16 + 4 * (lane >> 4);
        if (n < a.N) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
              const f32x4 c = acc4[i][sx][cb];
              const int64_t m = ((int64_t)img * a.H + y0 + i) * a.W + x0 + sx * 16 + (lane & 15);
              *(float4*)(slab + m * a.N + n) = make_float4(c[0], c[1], c[2], c[3]);
            }
        }
      }
      STAMP(3);
      return;
    }
  }
  if constexpr (SPL) {
    // ---- split-output epilogue (precise / mixed mode): [C/32][hi 32 | lo 32] f16 pairs per pixel.  The tile's values go through LDS as
    // fp32 (the same 4 bytes per element as the pair) in TWO passes of BN / 2 channels -- the waves owning them write their accumulators
    // (+ bias, activation), then every wave streams pixel rows out: a lane reads 8 channels (32 B), adds the residual's hi + lo (two 16-byte
    // loads issued before the pass), accumulates the output statistics, splits and writes 16 B of high parts and 16 B of low parts; the 4
    // lanes of a 32-channel group complete a 128-byte line with two back-to-back stores.  (Round 2 wrote 8-byte pieces straight from the
    // accumulators: 512x512 layers ran at 1.4-1.7 TB/s of algorithmic traffic, latency-bound on partial lines.)
    static_assert(MF16, "split outputs: 16x16x32 tiles");
    constexpr int HB = BN / 2, HW_ = NW / 2;             // channels, owning waves per pass
    constexpr int LPR = HB / 8;                          // lanes per pixel row of a pass (16 bytes of hi + 16 of lo each): 16 / 8
    constexpr int PPI = 64 / LPR;                        // pixels per store instruction: 4 / 8
    constexpr int PXW = NPX / NW;                        // pixels written out by a wave: 32 / 64
    constexpr int NWI = PXW / PPI;                       // iterations per wave and pass (8)
    char* const img_ = smem;
    float* const stat = (float*)(smem + NPX * FROW);     // [NW][HB][2] (sum, sumsq) partials per write-out wave, summed in a fixed order
    const int q = lane % LPR, psub = lane / LPR, cl0 = q * 8;
    constexpr int RD = WD_RD;                            // residual prefetch depth (iterations): 8 registers each, the accumulators are still live
    // per-image buffer resources; lane part of an address in the vector offset, the instruction's (wave-uniform) part in the scalar offset
    // (see the 16-bit epilogue below)
    const int rup = a.res_up ? 1 : 0, rsz = a.res_f32 ? 4 : 2;
    const int Hr = a.H >> rup, Wr = a.W >> rup;
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(a.R ? (const char*)a.R + (int64_t)img * Hr * Wr * a.ldr * rsz : nullptr, a.R ? (int64_t)Hr * Wr * a.ldr * rsz : 0);
    const __amdgpu_buffer_rsrc_t rs_d = make_rsrc((u16*)a.D + (int64_t)img * a.H * a.W * a.ldd, (int64_t)a.H * a.W * a.ldd * 2);
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      const int nq = n0 + hp * HB + cl0;                 // this lane's 8 logical output channels (inside one 32-group)
      const int po = split_off(nq, 32);                  // their high parts inside a pixel row; low parts 32 elements further
      uint4 rh[RD], rl[RD];
      const uint32_t rvo = (uint32_t)((psub >> rup) * a.ldr + (a.res_f32 ? nq : po)) * (uint32_t)rsz;
      const uint32_t rlo = a.res_f32 ? 16u : 64u;        // second half: the next 4 fp32 values, or the low parts 32 elements on
      const uint32_t dvo = (uint32_t)(psub * a.ldd + po) * 2u;
      auto load_res = [&](int t) {                       // a split residual (16 B of high + 16 B of low parts) or an fp32 one (32 B): the same bytes
        const int pu = wid * PXW + t * PPI;              // the instruction's first pixel (PPI = 4 / 8 divides 32: pu & 31 is even)
        const int y = y0 + (pu >> 5), x = x0 + (pu & 31);
        const uint32_t so = (uint32_t)(((y >> rup) * Wr + (x >> rup)) * a.ldr) * (uint32_t)rsz;
        rh[t % RD] = buf_load16_nt(rs_r, rvo, so); rl[t % RD] = buf_load16_nt(rs_r, rvo + rlo, so);
      };
      if (a.R) {
#pragma unroll
        for (int t = 0; t < RD; ++t) load_res(t);        // the first residual loads fly while the accumulators are staged
      }
      if (wn / HW_ == hp) {                              // wave-uniform: this wave's 32 channels belong to the pass
        const int wl = wn - hp * HW_;
        act_switch(a.act, [&](auto act_c) __attribute__((always_inline)) {
          constexpr int ACT = decltype(act_c)::value;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const int cl = wl * 32 + cb * 16 + 4 * (lane >> 4);
            const float4 b = *(const float4*)(bsm + hp * HB + cl);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
              for (int sx = 0; sx < 2; ++sx) {
                const f32x4 c = acc4[i][sx][cb];
                float4 v = make_float4(c[0] * a.alpha + b.x, c[1] * a.alpha + b.y, c[2] * a.alpha + b.z, c[3] * a.alpha + b.w);
                v.x = act_apply(v.x, ACT); v.y = act_apply(v.y, ACT); v.z = act_apply(v.z, ACT); v.w = act_apply(v.w, ACT);
                *(float4*)(img_ + (i * 32 + sx * 16 + (lane & 15)) * FROW + cl * 4) = v;
              }
          }
        });
      }
      __syncthreads();
      if (hp == 0) STAMP(6);
      float cs[16];                                      // [0..7] sums, [8..15] sums of squares of this lane's 8 channels
#pragma unroll
      for (int e = 0; e < 16; ++e) cs[e] = 0.f;
#pragma unroll
      for (int t = 0; t < NWI; ++t) {
        const int p = wid * PXW + t * PPI + psub;
        float f[8];
        *(float4*)f = *(const float4*)(img_ + p * FROW + cl0 * 4);
        *(float4*)(f + 4) = *(const float4*)(img_ + p * FROW + cl0 * 4 + 16);
        if (a.R) {
          if (a.res_f32) {
            const uint4 u0 = rh[t % RD], u1 = rl[t % RD];
            f[0] += __uint_as_float(u0.x); f[1] += __uint_as_float(u0.y); f[2] += __uint_as_float(u0.z); f[3] += __uint_as_float(u0.w);
            f[4] += __uint_as_float(u1.x); f[5] += __uint_as_float(u1.y); f[6] += __uint_as_float(u1.z); f[7] += __uint_as_float(u1.w);
          } else {
            float r0[8], r1[8];
            unpack8<F16>(rh[t % RD], r0);
            unpack8<F16>(rl[t % RD], r1);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += r0[e] + r1[e];
          }
          if (t + RD < NWI) load_res(t + RD);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { cs[e] += f[e]; cs[8 + e] += f[e] * f[e]; }
        const uint4 vh = pack8<F16>(f);
        float hf[8], lo[8];
        unpack8<F16>(vh, hf);
#pragma unroll
        for (int e = 0; e < 8; ++e) lo[e] = f[e] - hf[e];
        const uint4 vl = pack8<F16>(lo);
        const int pu = wid * PXW + t * PPI;
        const uint32_t so = (uint32_t)(((y0 + (pu >> 5)) * a.W + x0 + (pu & 31)) * a.ldd) * 2u;
        buf_store16_nt(vh, rs_d, dvo, so);
        buf_store16_nt(vl, rs_d, dvo + 64u, so);
      }
      if (a.stats) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          cs[e] += __shfl_xor(cs[e], 32);
          cs[e] += __shfl_xor(cs[e], 16);
          if (LPR == 8) cs[e] += __shfl_xor(cs[e], 8);
        }
        if (lane < LPR) {
          float* const slot = stat + (wid * HB + cl0) * 2;
#pragma unroll
          for (int e = 0; e < 8; ++e) { slot[2 * e] = cs[e]; slot[2 * e + 1] = cs[8 + e]; }
        }
      }
      __syncthreads();                                   // every wave is done with the image (next pass overwrites it); statistic slots complete
      if (hp == 0) STAMP(7);
      if (a.stats) {
        float* o = a.stats + (((int64_t)img * a.stats_p + ty * tiles_x + tx) * a.N + n0 + hp * HB) * 2;
        for (int c = tid; c < 2 * HB; c += NT) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) v += stat[w * 2 * HB + c];
          o[c] = v;
        }
      }
    }
    STAMP(3);
    return;
  }
  // ---- epilogue: the whole tile goes through LDS once and leaves as full pixel rows (BN x 2 B contiguous) ----
  char* const stg = smem;
  float* const stat = (float*)(smem + NPX * SROW);     // [NW][BN][2] (sum, sumsq) partials per write-out wave, summed in a fixed order
  constexpr int LPR = BN / 8;                          // lanes per output pixel row (16 bytes each): 32 / 16
  constexpr int PPI = 64 / LPR;                        // pixels per store instruction: 2 / 4
  constexpr int PXW = NPX / NW;                        // pixels written out by a wave: 32 / 64
  constexpr int NWI = PXW / PPI;                       // store instructions per wave (16)
  const int q = lane % LPR, psub = lane / LPR;
  const int cl0 = q * 8;
  const bool col_live = n0 + cl0 < a.N;                // this lane's 8 output channels exist (Cout tail of the last tile)
  // the output is written, and the residual read, exactly once by this kernel: nontemporal hint (-0.35 % on the 512x512 step, neutral on
  // StableDiffusion's small maps; same-box A/B).  A run-time choice by output size cost registers the main loop does not have (+1.4 %).
  // Addresses: per-image buffer resources, the lane's part of the offset (its pixel inside the instruction's PPI, its 8 channels) in the
  // vector offset and the instruction's part (row, first pixel: wave-uniform) in the SCALAR offset -- the 64-bit per-lane products of the
  // pointer form (two v_mul_lo + v_mad_u64 per access, quarter rate) were ~2.5 of a 256-channel tile's 8.8 us of epilogue.  Dead lanes
  // (Cout tail) and an absent residual are out-of-range offsets / an empty resource: zeros, no traffic, no branch.
  const int rup = a.res_up ? 1 : 0;
  const int Hr = a.H >> rup, Wr = a.W >> rup;
  const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(a.R ? (const u16*)a.R + (int64_t)img * Hr * Wr * a.ldr : nullptr,
                                                 a.R ? ((int64_t)(Hr * Wr - 1) * a.ldr + a.N) * 2 : 0);
  const __amdgpu_buffer_rsrc_t rs_d = make_rsrc((u16*)a.D + (int64_t)img * a.H * a.W * a.ldd, ((int64_t)(a.H * a.W - 1) * a.ldd + a.N) * 2);
  const uint32_t rvo = col_live ? (uint32_t)((psub >> rup) * a.ldr + n0 + cl0) * 2u : PMI_BUF_OOB;
  const uint32_t dvo = col_live ? (uint32_t)(psub * a.ldd + n0 + cl0) * 2u : PMI_BUF_OOB;
  uint4 rres[NWI];
#pragma unroll
  for (int t = 0; t < NWI; ++t) {                      // residual loads fly while the accumulators are staged
    const int pu = wid * PXW + t * PPI;                // the instruction's first pixel (wave-uniform; PPI divides 32, so pu & 31 is even for PPI >= 2)
    const int y = y0 + (pu >> 5), x = x0 + (pu & 31);
    rres[t] = buf_load16_nt(rs_r, rvo, (uint32_t)(((y >> rup) * Wr + (x >> rup)) * a.ldr) * 2u);
  }
  // (no barrier here: the bias table is the prologue's, and the main loop's last barrier already freed the patch buffers -- what a slower
  // wave still reads from them is a prefetch past the last chunk that nothing uses)
  // bias values of this lane's columns, read once (a read between the staging writes cannot be hoisted by the compiler: same LDS)
  act_switch(a.act, [&](auto act_c) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_c)::value;
    if constexpr (MF16) {
      float4 bb[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) bb[cb] = *(const float4*)(bsm + wn * 32 + cb * 16 + 4 * (lane >> 4));
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const int cl = wn * 32 + cb * 16 + 4 * (lane >> 4);
            const float4 b = bb[cb];
            const f32x4 c = acc4[i][sx][cb];
            float v[4] = {c[0] * a.alpha + b.x, c[1] * a.alpha + b.y, c[2] * a.alpha + b.z, c[3] * a.alpha + b.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ACT);
            *(uint2*)(stg + (i * 32 + sx * 16 + (lane & 15)) * SROW + cl * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
          }
    } else {
      float4 bb[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bb[g] = *(const float4*)(bsm + wn * 32 + 8 * g + 4 * lhi);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cl = wn * 32 + 8 * g + 4 * lhi;
          const float4 b = bb[g];
          float v[4] = {acc[i][4 * g] * a.alpha + b.x, acc[i][4 * g + 1] * a.alpha + b.y,
                        acc[i][4 * g + 2] * a.alpha + b.z, acc[i][4 * g + 3] * a.alpha + b.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ACT);
          *(uint2*)(stg + (i * 32 + l31) * SROW + cl * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
        }
    }
  });
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
    const int pu = wid * PXW + t * PPI;
    buf_store16_nt(v, rs_d, dvo, (uint32_t)(((y0 + (pu >> 5)) * a.W + x0 + (pu & 31)) * a.ldd) * 2u);
  }
  STAMP(7);
  if (a.stats) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      cs[e] += __shfl_xor(cs[e], 32);
      if (LPR == 16) cs[e] += __shfl_xor(cs[e], 16);
    }
    if (lane < LPR) {
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
      if (n0 + (c >> 1) < a.N) o[c] = v;
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
  if (a.split_out) {                                   // precise / mixed mode (hi + lo output): configs 6 / 7 only, f16 arithmetic
    const dim3 g7(nimg * (a.H / 8) * (a.W / 32) * (a.N / 128)), g6(nimg * (a.H / 8) * (a.W / 32) * (a.N / 256));
    if (cfg != 6 && cfg != 7 && !(cfg == 8 && PRO == 0)) return PMI_ERR_ARG;
    if constexpr (PRO == 0) {
      if (a.split_in == 2) return PMI_ERR_ARG;
      if (cfg == 8) hipLaunchKernelGGL((conv3x3_wd_kernel<T, 0, true, 4, 32, true, false, true>), g7, dim3(256), 0, s, a);      // first convolution, split output
      else if (cfg == 7) hipLaunchKernelGGL((conv3x3_wd_kernel<T, 0, true, 4, 32, true>), g7, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((conv3x3_wd_kernel<T, 0, true, 8, 64, true>), g6, dim3(512), 0, s, a);
    } else if constexpr (PRO == 1 + PMI_ACT_SILU && std::is_same<T, F16>::value) {
      // fused GroupNorm-apply + SiLU over a split input: the doubled operand (split_in 1) or the single operand (split_in 2)
      if (a.split_in != 1 && a.split_in != 2) return PMI_ERR_ARG;
      if (cfg == 7) {
        if (a.split_in == 1) hipLaunchKernelGGL((conv3x3_wd_kernel<T, PRO, true, 4, 32, true, false, false, 1>), g7, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wd_kernel<T, PRO, true, 4, 32, true, false, false, 2>), g7, dim3(256), 0, s, a);
      } else {
        if (a.split_in == 1) hipLaunchKernelGGL((conv3x3_wd_kernel<T, PRO, true, 8, 64, true, false, false, 1>), g6, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wd_kernel<T, PRO, true, 8, 64, true, false, false, 2>), g6, dim3(512), 0, s, a);
      }
    } else {
      return PMI_ERR_ARG;
    }
    PMI_CHECK_LAUNCH();
    return PMI_OK;
  }
  if (a.split_in) return PMI_ERR_ARG;
  if (cfg == 8) {                                      // at most 32 input channels (first convolution): PRO == 0 only
    if constexpr (PRO == 0) {
      hipLaunchKernelGGL((conv3x3_wd_kernel<T, 0, true, 4, 32, false, false, true>), dim3(nimg * (a.H / 8) * (a.W / 32) * ((a.N + 127) / 128)), dim3(256), 0, s, a);
      PMI_CHECK_LAUNCH();
      return PMI_OK;
    }
    return PMI_ERR_ARG;
  }
  if (cfg == 7) {                                      // 128-channel tiles, two 4-wave workgroups per CU
    const int t7 = nimg * (a.H / 8) * (a.W / 32) * ((a.N + 127) / 128);
    if (a.splitk > 1) hipLaunchKernelGGL((conv3x3_wd_kernel<T, PRO, true, 4, 32, false, true>), dim3(t7, 1, a.splitk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv3x3_wd_kernel<T, PRO, true, 4, 32>), dim3(t7), dim3(256), 0, s, a);
    PMI_CHECK_LAUNCH();
    return PMI_OK;
  }
  const int tiles = nimg * (a.H / 8) * (a.W / 32) * (a.N / 256);
  if (cfg == 6 && a.splitk > 1) hipLaunchKernelGGL((conv3x3_wd_kernel<T, PRO, true, 8, 64, false, true>), dim3(tiles, 1, a.splitk), dim3(512), 0, s, a);
  else if (cfg == 6) hipLaunchKernelGGL((conv3x3_wd_kernel<T, PRO, true>), dim3(tiles), dim3(512), 0, s, a);    // v_mfma_f32_16x16x32
  else hipLaunchKernelGGL((conv3x3_wd_kernel<T, PRO, false>), dim3(tiles), dim3(512), 0, s, a);             // config 4: 32x32x16
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

// Split-K factor for the weights-direct configs 6 / 7 (1 = none): output tiles alone leave most CUs idle on 32x32 maps at batch 8
// (128 workgroups of 128 channels for a 512-channel layer); the reduction is split so that each part keeps >= 256 input channels.
static int g_wd_splitk = 1;        // A/B switch: pmi_set_option(10, 0 / 1)
void pmi_conv3x3_wd_splitk_enable(int v) { g_wd_splitk = v; }
int pmi_conv3x3_wd_splitk(const pmi_igemm_args* a, int cfg) {
  if (!g_wd_splitk || (cfg != 6 && cfg != 7) || a->split_out || a->split_in || a->out_f32 || (a->R && a->res_f32) || (a->N & 3)) return 1;
  const int ck = cfg == 6 ? 64 : 32, bn = cfg == 6 ? 256 : 128;
  const int nchunks = (a->C0 + a->C1) / ck;
  const long wgs = (long)(a->M / (a->H * a->W)) * (a->H / 8) * (a->W / 32) * ((a->N + bn - 1) / bn);
  if (wgs > 128) return 1;          // same-box A/B: 128 workgroups (512-channel layers, 32x32 x 8): c5 43.06 -> 42.82 ms; 160 (640 channels, c4): 15.32 -> 15.46, not split
  int best = 1;
  for (int s = 2; s <= 4; s *= 2) {
    if (nchunks % s) continue;
    const int per = nchunks / s;
    if (per * ck < 256 || (cfg == 7 && (per & 1)) || wgs * s > 512) continue;
    best = s;
  }
  return best;
}

// (Round 3: a start-time stagger of the first round's workgroups -- phase x chunks x ticks, so that the CUs' epilogue bursts spread over time --
// was measured on the c5 step, mixed: 61.2 -> 62.4 / 63.9 / 65.6 ms at 3 / 6 / 10 us per chunk and phase, bf16: 42.3 -> 43.0 / 43.2: the
// offsets only add idle time, the epilogues do not get faster when fewer CUs run them at once.  Removed.)
// (Also measured and removed in round 3: split-output tiles starting their accumulators from the residual, loaded by each lane for its own
// accumulator elements before the prologue, so that the epilogue's write-out is stores only.  The epilogue went 33.6 -> 20.9 us per tile but the
// 8-byte accumulator-layout loads took 22 us of prologue: c5 mixed step 63.0 -> 64.6 ms on one box.)
int pmi_conv3x3_wd_launch(const pmi_igemm_args* a, int cfg, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  return a->dtype == PMI_DT_BF16 ? launch_t<BF16>(*a, s, cfg) : launch_t<F16>(*a, s, cfg);
}
