// gemm256q.h -- k_gemm256q: the 256 x 256 x 64 GEMM tile with FOUR waves of 128 x 128 (one per SIMD, accumulators in the
// AGPR half) instead of k_gemm256's eight of 128 x 64.  Built in round 5, bit-identical to k_gemm256 / k_gemm in every
// epilogue, MEASURED SLOWER (profiles/archive/r05_e_gemm256q.log: FFN-up 26.5 against 23.8 us, FFN-down 52.6 against 47.4 us at
// M = 8064; 15-25 % at M = 32,256) and therefore not part of the library: it is compiled only into tools/gemm_bench
// (mode 3), which defines QV_GEMM_Q_VARIANT to this file before including csrc/qv_gemm256.hip.
// What the ablations say (QV_Q_ABL, same log): without its buffer loads the kernel is no faster, without loads, stage stores
// and barriers FFN-up is still 24.5 us -- a lone wave per SIMD with nothing but fragment reads and MFMAs does not beat the
// eight-wave kernel WITH all of its staging, so the address-path queueing the variant was built to remove is not what
// bounds the K loop.
#pragma once

// ---------------------------------------------------------------------------------------------------------------------
// k_gemm256q (round 5): the same 256 x 256 x 64 tile with FOUR waves, one per SIMD, each owning a 128 x 128 sub-tile
// (4 x 4 accumulators of 32 x 32 = 256 registers, which the register allocator places in the AGPR half of the 512 a lone
// wave per SIMD may use).  Why: in k_gemm256 eight waves queue their buffer loads at the same points of the K-tile and a
// wave that waits for the CU's single address path cannot issue its MFMAs meanwhile, so the K-tile costs about (matrix-
// pipe time + address-path time) (DESIGN.md "GEMM, round 2b"); loader waves cannot be split off because a kernel has ONE
// register allocation for all its waves.  Here a wave issues ONE store + re-request pair behind every fourth MFMA -- 16
// pairs for 64 MFMAs per K-tile, each short enough to hide behind the MFMA in flight -- only four waves contend for the
// address path, and the fragment traffic drops from 192 to 128 KB per K-tile (a wave reads 4 A + 4 W fragments per 16
// MFMAs instead of 4 + 2 per 8).  LDS image, swizzle, MFMA instruction and the K order of every accumulator are those of
// k_gemm256 / k_gemm, so every output is bit-identical to theirs (tools/gemm_bench mode 3, tests/test_gpu_gemm256.py).
// Loads are inline asm with hand-counted vmcnt like the other kernels (tests/test_loader_hazards.py checks them): the
// outstanding queue is always [pieces of tile kt + 1 not yet stored..., pieces of tile kt + 2 already re-requested...],
// NSW long in the steady state, so the piece about to be stored is complete at vmcnt(NSW - 1).
namespace {
template <int I> using ic_t = std::integral_constant<int, I>;
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(ic_t<I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
}  // namespace

template <int EPI, int WQ>
__global__ __launch_bounds__(256, 1) void k_gemm256q(GemmArgs g) {
    static_assert(WQ == 0 || WQ == 4 || WQ == 8, "f16 activations only");
    constexpr bool W4 = WQ == 4, W8 = WQ == 8;
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = W4 ? BN * BK / 2 : W8 ? BN * BK : BN * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NPA = 8;                        // 1 KB pieces of the A tile per wave per K-tile
    constexpr int NBP = W4 ? 2 : W8 ? 4 : 8;      // ... of the W tile
    constexpr int NSW = NPA + NBP;                // store + re-request pairs per wave per K-tile
    constexpr int WGAP = 8 / NBP;                 // W piece p is handled in slot 8 + p * WGAP of the 16 slots of a K-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int gx = gridDim.x, nwg = gridDim.x * gridDim.y;
    int wg = blockIdx.y * gx + blockIdx.x;
    {   // XCD-aware tile order, as in k_gemm256
        int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (wg / gx) * BM, n0 = (wg % gx) * BN;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // ------------------------------------------------------------------ staging ----------
    // piece q of this wave: tile rows wave*64 + q*8 .. +7 of A (and of an f16 W), 128 B per row; lane -> (row lane>>3,
    // 16-byte chunk lane&7); chunk c of LDS row r sits at c ^ ((r >> 1) & 7)
    const int nk = g.K / BK;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, (int)((size_t)g.M * g.lda * 2), 0x00020000);
    const void *bbase = W4 ? (const void *)g.Wq : W8 ? (const void *)g.W8 : (const void *)g.W;
    const size_t bbytes = W4 ? (size_t)g.N * g.K / 2 : W8 ? (size_t)g.N * g.K : (size_t)g.N * g.ldw * 2;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)bbase, 0, (int)bbytes, 0x00020000);
    constexpr int stepB = W4 ? 2048 : W8 ? 4096 : BK * 2;   // bytes between consecutive K-tiles of a W piece
    unsigned offA[NPA], offB[NBP];
    int dstA[NPA], dstB[NBP];
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
        const int row = wave * 64 + q * 8 + (lane >> 3), c = lane & 7;
        int grow = m0 + row;
        grow = grow < g.M ? grow : g.M - 1;   // rows past M repeat the last one; their outputs are never stored
        offA[q] = (unsigned)(((size_t)grow * g.lda + c * 8) * 2);
        dstA[q] = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        if (!W4 && !W8) {
            offB[q] = (unsigned)(((size_t)(n0 + row) * g.ldw + c * 8) * 2);
            dstB[q] = A_BYTES + dstA[q];
        }
    }
    if (W4 || W8) {
        // 64 x 64 code tiles, 2 KB (int4) / 4 KB (int8) contiguous and already in the LDS order: the tile's 8 / 16 pieces
        // of 1 KB, wave w takes pieces w * NBP .. w * NBP + NBP - 1
        constexpr int PPT = W4 ? 2 : 4;           // pieces per 64-column code tile
#pragma unroll
        for (int q = 0; q < ((W4 || W8) ? NBP : 0); ++q) {
            const int pc = wave * NBP + q;
            offB[q] = (unsigned)(((size_t)((n0 >> 6) + pc / PPT) * nk) * (PPT * 1024) + (pc % PPT) * 1024 + lane * 16);
            dstB[q] = A_BYTES + pc * 1024 + lane * 16;
        }
    }
    u32x4 ra[NPA], rb[NBP];
    auto fetch_all = [&](int kt) {
#pragma unroll
        for (int q = 0; q < NPA; ++q)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[q]) : "v"(offA[q]), "s"(rsA), "s"(kt * (BK * 2)) : "memory");
#pragma unroll
        for (int q = 0; q < NBP; ++q)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[q]) : "v"(offB[q]), "s"(rsB), "s"(kt * stepB) : "memory");
    };
    auto put_all = [&](int stage) {
        unsigned char *st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < NPA; ++q) *(u32x4 *)(st + dstA[q]) = ra[q];
#pragma unroll
        for (int q = 0; q < NBP; ++q) *(u32x4 *)(st + dstB[q]) = rb[q];
    };

    // W4: this tile's {scale, 1024 + zero point} pairs [K/128][BN] stay in LDS behind the stage ring for the whole K loop
    h2_t *sS = (h2_t *)(smem + 2 * STAGE_BYTES);
    if (W4) {
        const int nkb = g.K >> 7;
        for (int idx = tid; idx < BN * nkb; idx += 256) {
            const int kb = idx / BN, n = idx - kb * BN;
            sS[idx] = ((const h2_t *)g.wscale)[(size_t)kb * g.N + n0 + n];
        }
    }

    fetch_all(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    put_all(0);
    if (nk > 1) fetch_all(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    struct Frag { half8 a[4]; half8 b[4]; uint32_t q4[4]; uint2 q8[4]; };
    const int l31 = lane & 31, hi = lane >> 5;
    auto rdB = [&](int stage, int ks, Frag &f) {
        const unsigned char *sB = smem + stage * STAGE_BYTES + A_BYTES;
        const int c = ks * 2 + hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wn * 128 + j * 32 + l31;
            if (W4) f.q4[j] = *(const uint32_t *)(sB + row * 32 + ((c ^ ((row >> 2) & 7)) << 2));
            else if (W8) f.q8[j] = *(const uint2 *)(sB + (row >> 6) * 4096 + (row & 63) * 64 + ((c ^ ((row >> 2) & 7)) << 3));
            else f.b[j] = *(const half8 *)(sB + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        }
    };
    auto rdA = [&](int stage, int ks, Frag &f) {
        const unsigned char *sA = smem + stage * STAGE_BYTES;
        const int c = ks * 2 + hi;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 128 + i * 32 + l31;
            f.a[i] = *(const half8 *)(sA + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        }
    };
    h2_t sc[4], zo[4];   // W4: this K-tile's scale / offset pairs of the wave's four column fragments

    // One K-tile = 16 slots (sub-step s = slot >> 2 of 16 k, A fragment i = slot & 3): 4 MFMAs (the four column fragments
    // against A fragment i), then the slot's store + re-request pair.  The fragments of sub-step s + 1 are read in slots
    // 0 (W) and 1 (A) of sub-step s into the other register set.
    auto ktile = [&](int kt, auto last_c, auto has2_c) {
        constexpr bool LAST = decltype(last_c)::value;
        constexpr bool has2 = decltype(has2_c)::value;      // tile kt + 2 exists: every stored piece is re-requested
        const int cur = kt & 1;
        unsigned char *nxt = smem + (cur ^ 1) * STAGE_BYTES;
        Frag f[2];
        if (W4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h2_t sz = sS[(kt >> 1) * BN + wn * 128 + j * 32 + l31];
                sc[j] = h2_t{sz[0], sz[0]};
                zo[j] = h2_t{sz[1], sz[1]};
            }
        }
        rdB(cur, 0, f[0]);
        rdA(cur, 0, f[0]);
        half8 bq[4];
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto tc) {
            constexpr int t = decltype(tc)::value, s = t >> 2, i = t & 3;
            Frag &fs = f[s & 1];
            if (s < 3 && i == 0) rdB(cur, s + 1, f[(s + 1) & 1]);
            if (s < 3 && i == 1) rdA(cur, s + 1, f[(s + 1) & 1]);
            if (i == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[j] = W4 ? dequant8(fs.q4[j], sc[j], zo[j]) : W8 ? dequant8_i8(fs.q8[j]) : fs.b[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)   // operands swapped (D^T = W A^T): a lane holds 4 CONSECUTIVE output columns per register quad
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[j], fs.a[i], acc[i][j], 0, 0, 0);
            if (!LAST) {
                constexpr bool IS_A = t < NPA;
                constexpr bool IS_W = t >= 8 && (t - 8) % WGAP == 0 && (t - 8) / WGAP < NBP;
                if constexpr (IS_A || IS_W) {
                    constexpr int P = IS_A ? t : NPA + (t - 8) / WGAP;       // position in the request order
#ifndef QV_Q_ABL
#define QV_Q_ABL 0   // dev ablations (tools/gemm_bench -DQV_Q_ABL=..): 1 no buffer loads in the loop, 2 no stage stores, 4 no barrier
#endif
                    if (has2 && !(QV_Q_ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSW - 1) : "memory");
                    else if (!(QV_Q_ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSW - 1 - P) : "memory");
                    if constexpr (IS_A) {
                        if (!(QV_Q_ABL & 2)) *(u32x4 *)(nxt + dstA[t]) = ra[t];
                        if (has2 && !(QV_Q_ABL & 1)) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[t]) : "v"(offA[t]), "s"(rsA), "s"((kt + 2) * (BK * 2)) : "memory");
                    } else {
                        constexpr int q = (t - 8) / WGAP;
                        if (!(QV_Q_ABL & 2)) *(u32x4 *)(nxt + dstB[q]) = rb[q];
                        if (has2 && !(QV_Q_ABL & 1)) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[q]) : "v"(offB[q]), "s"(rsB), "s"((kt + 2) * stepB) : "memory");
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // stage kt + 1 written (own ds_writes retired) and stage kt read by every wave
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(QV_Q_ABL & 4)) __builtin_amdgcn_s_barrier();
    };
    for (int kt = 0; kt + 2 < nk; ++kt) ktile(kt, std::false_type{}, std::true_type{});
    if (nk >= 2) ktile(nk - 2, std::false_type{}, std::false_type{});
    ktile(nk - 1, std::true_type{}, std::false_type{});

    // accumulator (i, j), register r: tile row = wm*128 + i*32 + (lane & 31),
    //   tile column = wn*128 + j*32 + 8*(r >> 2) + 4*(lane >> 5) + (r & 3)
    const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc((void *)g.bias, 0, g.N * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_scl = __builtin_amdgcn_make_buffer_rsrc((void *)(W8 ? g.w8scale : g.bias), 0, g.N * 4, 0x00020000);
    auto ldf4 = [&](const __amdgpu_buffer_rsrc_t &rs, int elem) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, elem * 4, 0, 0));
    };
    const int nb = n0 + wn * 128;         // first tile column of this wave
    const int mb = m0 + wm * 128;         // first row of this wave
    // ------------------------------------------------------------------ epilogue ----------
    // wave-private like k_gemm256's: 32 rows at a time through the wave's own 32 KB slice of the idle stage ring.  The
    // bias (and W8A16 scale) vectors of a 32-column fragment are fetched when its turn comes: 16 column quads x 2 tables
    // held at once would not leave room beside the 256 accumulator registers.
    unsigned char *sW = smem + wave * 32768;
    auto col_tab = [&](int j, f32x4 b4[4], f32x4 s4[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b4[q] = ldf4(rs_bias, nb + j * 32 + 8 * q + 4 * hi);
            if (W8) s4[q] = ldf4(rs_scl, nb + j * 32 + 8 * q + 4 * hi);
        }
    };

    if (EPI == EPI_QKV && n0 >= 2 * QV_D) {
        // V tile: stored TRANSPOSED, Vt[b][h*64+d][t] (see k_gemm256); this wave's 128 columns are two heads' worth of d
        constexpr int LDV = 64 + 2;   // halves per d row (odd dword pitch)
        half_t *sT = (half_t *)sW;
        half_t *vt = (half_t *)g.out2;
        const int fp = lane & 31, dsub = lane >> 5;
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 b4[4], s4[4];
                    col_tab(2 * jh + j, b4, s4);
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                sT[(j * 32 + 8 * q + 4 * hi + e) * LDV + ii * 32 + l31] = (half_t)(acc[hh * 2 + ii][2 * jh + j][q * 4 + e] + b4[q][e]);
                }
                const int r0 = mb + hh * 64 + 2 * fp, r1 = r0 + 1;
                const int bt0 = r0 < g.M ? g.row_map[r0] : -1, bt1 = r1 < g.M ? g.row_map[r1] : -1;
                const bool pair = bt0 >= 0 && bt1 == bt0 + 1 && (bt0 & 1) == 0;   // same utterance, even frame: one 4-byte store
                for (int dd = 0; dd < 32; ++dd) {
                    const int d = dd * 2 + dsub;
                    const size_t drow = (size_t)(nb + jh * 64 - 2 * QV_D + d);
                    const half_t v0 = sT[d * LDV + 2 * fp], v1 = sT[d * LDV + 2 * fp + 1];
                    if (pair) {
                        h2_t v = {v0, v1};
                        *(h2_t *)(vt + ((size_t)(bt0 >> 16) * QV_D + drow) * g.t_pad + (bt0 & 0xFFFF)) = v;
                    } else {
                        if (bt0 >= 0) vt[((size_t)(bt0 >> 16) * QV_D + drow) * g.t_pad + (bt0 & 0xFFFF)] = v0;
                        if (bt1 >= 0) vt[((size_t)(bt1 >> 16) * QV_D + drow) * g.t_pad + (bt1 & 0xFFFF)] = v1;
                    }
                }
            }
        }
        return;
    }

    if (out_is_f32(EPI)) {
        constexpr int LDT = 128 + 4;   // floats per staged row
        float *sO = (float *)sW;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.M * g.ldo * 4), 0x00020000);
        const int rr = lane >> 5, cc = (lane & 31) * 4;   // read-back: 2 rows x 512 B per instruction
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 old[16];
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int r = mb + i * 32 + k * 2 + rr;
                    if (r < g.M) old[k] = ldf4(rs_out, r * g.ldo + nb + cc);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 b4[4], s4[4];
                col_tab(j, b4, s4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][j][q * 4 + e];
                        if (W8) x *= s4[q][e];
                        v[e] = g.alpha * (x + b4[q][e]);
                    }
                    *(f32x4 *)(sO + l31 * LDT + j * 32 + 8 * q + 4 * hi) = v;
                }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int r = mb + i * 32 + k * 2 + rr;
                f32x4 v = *(const f32x4 *)(sO + (k * 2 + rr) * LDT + cc);
                if (r >= g.M) { asm volatile("" ::"v"(v)); continue; }
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += old[k][e];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, (r * g.ldo + nb + cc) * 4, 0, 0);
            }
        }
        return;
    }

    if (EPI == EPI_GLU) {
        // W rows interleaved in 32-channel groups, [value(32) | gate(32)] per 64 columns: the wave's four accumulator
        // columns are two value / gate pairs -> 64 output channels
        constexpr int LDT = 64 + 8;
        half_t *sO = (half_t *)sW;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.M * g.ldo * 2), 0x00020000);
        const int rr = lane >> 3, cc = (lane & 7) * 8;    // read-back: 8 rows x 128 B per instruction
        const int nbo = n0 / 2 + wn * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4 ba[4], bg[4], sa[4], sg4[4];
                col_tab(2 * pr, ba, sa);
                col_tab(2 * pr + 1, bg, sg4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4 o;
                    f32x4 av, gv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        av[e] = acc[i][2 * pr][q * 4 + e];
                        gv[e] = acc[i][2 * pr + 1][q * 4 + e];
                        if (W8) { av[e] *= sa[q][e]; gv[e] *= sg4[q][e]; }
                        av[e] += ba[q][e];
                        gv[e] += bg[q][e];
                    }
                    const f32x4 sg = sigmoid4(gv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)(av[e] * sg[e]);
                    *(half4 *)(sO + l31 * LDT + pr * 32 + 8 * q + 4 * hi) = o;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = mb + i * 32 + k * 8 + rr;
                const u32x4 v = *(const u32x4 *)(sO + (k * 8 + rr) * LDT + cc);
                if (r < g.M) __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, (r * g.ldo + nbo + cc) * 2, 0, 0);
            }
        }
        return;
    }

    {
        constexpr int LDT = 128 + 8;   // halves per staged row
        half_t *sO = (half_t *)sW;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.M * g.ldo * 2), 0x00020000);
        const int rr = lane >> 4, cc = (lane & 15) * 8;    // read-back: 4 rows x 256 B per instruction
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 b4[4], s4[4];
                col_tab(j, b4, s4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4 o;
                    f32x4 xv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = acc[i][j][q * 4 + e] + b4[q][e];
                    if (EPI == EPI_F16_SWISH) xv = swish4(xv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = xv[e];
                        if (EPI == EPI_F16_RELU) x = x > 0.f ? x : 0.f;
                        o[e] = (half_t)x;
                    }
                    *(half4 *)(sO + l31 * LDT + j * 32 + 8 * q + 4 * hi) = o;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = mb + i * 32 + k * 4 + rr;
                const u32x4 v = *(const u32x4 *)(sO + (k * 4 + rr) * LDT + cc);
                if (r < g.M) __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, (r * g.ldo + nb + cc) * 2, 0, 0);
            }
        }
    }
}

// 0 = k_gemm256 (eight waves), 1 = k_gemm256q (four waves) for the f16-activation variants; -1 = QVERSE_GEMM_Q / default
static int g_gemm_q = -1;
void qv_gemm_set_q(int mode) { g_gemm_q = mode; }
static bool gemm_q() {
    static const int env = [] { const char *e = getenv("QVERSE_GEMM_Q"); return e ? atoi(e) : 0; }();
    return (g_gemm_q >= 0 ? g_gemm_q : env) != 0;
}

