// NOT BUILT.  The forward counterpart of uniir_amd/csrc/attention_pair.hip (persistent 8-wave workgroup, K / V double buffered over
// heads by LDS-DMA, a pair of query tiles per wave, coalesced stores), as it was measured in round 4 and removed from the library.
// Correct (pair rows bitwise equal to attention.hip's forward, odd-tile rows within one bf16 ulp), every operand fetched once --
// and not faster: 257 tokens x 16 heads x 1024 items 0.70-0.72 ms vs 0.66-0.69 ms for the general kernel with the same coalesced
// stores; 197 tokens 0.49 vs 0.46-0.49.  Why (rocprofv3 --pmc, tools/r4/attn_pmc.sh): the forward is bound by the NUMBER of vector
// instructions a SIMD issues (213 per 32-key block of two query tiles, 16 of them MFMAs; ~4.5 cycles each, MFMAs 16, and the costs
// ADD on a SIMD), and pairing tiles halves only the MFMA and LDS instruction counts per flop, not the ~8 softmax instructions per
// logit.  To paste back: include after ApDma / ApBase in attention_pair.hip and call launch_attn_fwd_pair from launch_attn_fwd.
// SECOND MEASUREMENT (round 4, after attention.hip's forward got the per-row deferred maximum: 0.61-0.62 ms): this kernel with the same
// softmax (ApFwdState::l4 as four packed partials, the lane-local threshold test for both tiles, one scalar fold, cross-lane maximum
// and rescale only for a flagged row -- pair rows still bitwise equal to the general kernel's) compiles to 53 + 16 vector
// instructions, 16 MFMAs, 12 LDS reads and 42 scalar instructions per 32-key block of two tiles -- per tile what the general kernel
// issues, minus six LDS reads -- and ran 0.652-0.706 ms at 257 x 16 x 1024 against 0.622 (197 tokens: 0.447 vs 0.447).  Splitting the
// block loop into a DMA and a no-DMA body (the backward's trick) made hipcc emit 86 vector instructions per block instead of 53.
// Not kept, again: two co-resident workgroups of the general kernel hide the per-head staging as well as the double buffering here.

// =================================================================================================================================
// Forward.  K, V of a head in LDS, double buffered over heads: the next head's slices arrive by DMA while this one computes.  A wave
// owns a pair of query tiles (lane = query column, S^T = K Q^T, O^T = V^T P^T with P^T straight from the accumulators, online softmax
// per tile exactly as in attention.hip -- the pair rows come out bitwise equal); the odd tile's key blocks are dealt to the waves, whose
// partial (max, sum, O^T) are combined behind the head's barrier.
// =================================================================================================================================
template <int QT>
struct ApFwdState {
    f32x4_t o[QT][4];
    float m[QT], l[QT];
    DEVINL void init() {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            m[qt] = -1e30f;
            l[qt] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
};
// the key blocks b_lo, b_lo + b_step, .. < nblk of QT query tiles; only the last block holds padded keys.  LDS reads one MFMA
// group ahead: [V^T fragments of block b]  S^T MFMAs  [K rows of the next block]  online softmax  P V MFMAs
template <int TP, int QT>
DEVINL void ap_fwd_blocks(unsigned lk, const ApOff& of, const bf16x8_t (&qf)[QT][2], int T, int b_lo, int nblk, int b_step,
                          ApFwdState<QT>& st, int lane, ApDma& dma) {
    constexpr int SB = TP * 128;
    const int g = lane >> 4;
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    ApBase ba;
    ba.init(lk, of);
    bf16x8_t RK[2][2];
    auto load_rows = [&](int b) {
        const unsigned blk = (unsigned)b * 4096u;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            RK[kt][0] = ap_rows(ba.r[0] + blk + kt * 2048);
            RK[kt][1] = ap_rows(ba.r[1] + blk + kt * 2048);
        }
    };
    if (b_lo < nblk) load_rows(b_lo);
    for (int b = b_lo; b < nblk; b += b_step) {
        const unsigned blk = (unsigned)b * 4096u;
        bf16x8_t TV[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) TV[dt] = ap_cols(ba.t[dt] + (blk + SB));
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t sx[2][QT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sx[kt][qt] = mfma16(RK[kt][0], qf[qt][0], zero4);
        __builtin_amdgcn_sched_barrier(0);       // (first k-steps, then second k-steps: see the backward's phase 1)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sx[kt][qt] = mfma16(RK[kt][1], qf[qt][1], sx[kt][qt]);
        __builtin_amdgcn_sched_barrier(0);
        dma.step(lane);
        if (b + b_step < nblk) load_rows(b + b_step);
        __builtin_amdgcn_sched_barrier(0);
        // acc[r] = S^T[key = 32 b + 16 kt + 4 g + r][q = lane column]
        const bool edge = b == nblk - 1;
        const int krem = T - (b * 32 + 4 * g);
        bf16x8_t pf[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (edge) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sx[kt][qt][r] = (kt * 16 + r < krem) ? sx[kt][qt][r] : -1e30f;
            }
            float mx = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sx[kt][qt][r]);
            mx = group_max(mx);
            const float m_new = fmaxf(st.m[qt], mx * SCALE_LOG2E);
            if (__any(m_new > st.m[qt])) {
                const float alpha = __builtin_amdgcn_exp2f(st.m[qt] - m_new);
                st.l[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) st.o[qt][dt] = st.o[qt][dt] * alpha;
                st.m[qt] = m_new;
            }
            float sum = 0.f;
            f32x4_t p[2];
            const f32x4_t sc4 = {SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E};
            const f32x4_t nm4 = {-st.m[qt], -st.m[qt], -st.m[qt], -st.m[qt]};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {       // masked logits (-1e30) underflow to exactly 0; key 0 is valid for every row
                const f32x4_t arg = __builtin_elementwise_fma(sx[kt][qt], sc4, nm4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[kt][r] = __builtin_amdgcn_exp2f(arg[r]);
                    sum += p[kt][r];               // (the order of attention.hip's row sum: pair rows stay bitwise equal)
                }
            }
            st.l[qt] += sum;
            pf[qt] = pack8(p[0], p[1]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) st.o[qt][dt] = mfma16(TV[dt], pf[qt], st.o[qt][dt]);
    }
}

#define AP_FPART 68       // floats per (wave, odd-tile row): m, l, 2 unused, O[64]

template <int TP>
__global__ __launch_bounds__(AP_THREADS, 2) void attn_fwd_pair_kernel(AttnArgs a, ApGeom gm) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SB = TP * 128;
    const int T = gm.T, H = a.H, nblk = gm.nblk, npair = gm.npair, nvl = gm.nvl;
    float* part = reinterpret_cast<float*>(lds + 4 * SB);       // [2 heads][8 waves][nvl][AP_FPART]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    ApOff of;
    of.init(lane);
    const unsigned aL = ap_lds_addr(lds);
    const int q0 = 32 * w, qL0 = 32 * npair;
    const int idle = AP_WAVES - npair;      // the odd tile's key blocks: as in the backward
    const int lb_lo = gm.left ? (idle > 0 ? (w >= npair ? w - npair : nblk) : (w < 4 ? w : nblk)) : nblk;
    const int lb_step = idle > 0 ? idle : 4;
    {   // rows T .. TP - 1 of the four slices stay zero (V's padded rows meet P = 0: they must be finite)
        const int npad = (TP - T) * 8;
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (int c = tid; c < 4 * npad; c += AP_THREADS) {
            const int sl = c / npad, r = c % npad;
            *reinterpret_cast<u32x4_t*>(lds + sl * SB + T * 128 + r * 16) = z;
        }
    }
    const unsigned kbytes = (unsigned)((long)(T - 1) * a.kv_ld * 2 + 128);
    int hd = blockIdx.x;
    if (hd >= gm.total_heads) return;
    bf16x8_t qf[2][2], qf1[1][2];
    {   // first head: its queries, K and V
        const int m = hd / H, h = hd % H;
        const unsigned short* qbase = a.q + (long)m * T * a.q_ld + h * ATT_D;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[0][s] = ap_frag_global(qbase, a.q_ld, q0, s, lane, T);
            qf[1][s] = ap_frag_global(qbase, a.q_ld, q0 + 16, s, lane, T);
            qf1[0][s] = ap_frag_global(qbase, a.q_ld, qL0, s, lane, T);
        }
        ap_wait_vm0();
#pragma unroll
        for (int s = 0; s < 2; ++s) { ap_pin(qf[0][s]); ap_pin(qf[1][s]); ap_pin(qf1[0][s]); }
        ap_stage2(ap_make_srd(a.k + (long)m * T * a.kv_ld + h * ATT_D, kbytes), a.kv_ld,
                  ap_make_srd(a.v + (long)m * T * a.kv_ld + h * ATT_D, kbytes), a.kv_ld, aL, aL + SB, T, w, lane);
        ap_wait_vm0();
        __syncthreads();
    }
    int it = 0;
    for (; hd < gm.total_heads; hd += gridDim.x, ++it) {
        const int m = hd / H, h = hd % H;
        const int nh = hd + gridDim.x;
        const bool more = nh < gm.total_heads;
        const int m2 = more ? nh / H : m, h2 = more ? nh % H : h;
        const unsigned cur = aL + (unsigned)(it & 1) * (2 * SB), nxt = aL + (unsigned)((it + 1) & 1) * (2 * SB);
        float* mypart = part + (it & 1) * (AP_WAVES * nvl * AP_FPART);
        AP_STAMP(0);
        // the queries requested at the end of the last head are complete (no DMA is in flight here)
        ap_wait_vm0();
#pragma unroll
        for (int s = 0; s < 2; ++s) { ap_pin(qf[0][s]); ap_pin(qf[1][s]); ap_pin(qf1[0][s]); }
        int tid1 = tid;
        asm volatile("" : "+v"(tid1));
        ApDma dma;     // the next head's K, V into the other buffer (released by the barrier that ended the head before this one)
        if (more)
            dma.start(ap_make_srd(a.k + (long)m2 * T * a.kv_ld + h2 * ATT_D, kbytes), a.kv_ld,
                      ap_make_srd(a.v + (long)m2 * T * a.kv_ld + h2 * ATT_D, kbytes), a.kv_ld, nxt, nxt + SB, T, w);
        else
            dma.idle();
        AP_STAMP(1);
        if (gm.left) {          // this wave's share of the odd tile: partial (max, sum, O^T) of its key blocks
            ApFwdState<1> s1;
            s1.init();
            ap_fwd_blocks<TP, 1>(cur, of, qf1, T, lb_lo, nblk, lb_step, s1, lane, dma);
            const float lsum = group_sum(s1.l[0]);
            const int li = lane & 15, g = lane >> 4;
            if (li < nvl) {
                float* p = mypart + (w * nvl + li) * AP_FPART;
                if (g == 0) {
                    p[0] = s1.m[0];
                    p[1] = lsum;
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(p + 4 + 16 * dt + 4 * g) = s1.o[0][dt];
            }
        }
        AP_STAMP(2);
        ApFwdState<2> st;
        st.init();
        if (w < npair) ap_fwd_blocks<TP, 2>(cur, of, qf, T, 0, nblk, 1, st, lane, dma);
        AP_STAMP(3);
        int tid2 = tid;
        asm volatile("" : "+v"(tid2));
        const int lane2 = tid2 & 63, li2 = lane2 & 15, g2 = lane2 >> 4;
        dma.drain(lane2);
        ap_wait_vm0();          // the next head's K / V pieces of this wave have landed
        AP_STAMP(4);
        if (w < npair) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int q = q0 + qt * 16 + li2;
                const float lsum = group_sum(st.l[qt]);
                att_store_tile(st.o[qt], 1.0f / lsum, a.out + ((long)m * T + q) * a.out_ld + h * ATT_D, q < T, g2);
                if (g2 == 0 && q < T) a.lse[((long)m * H + h) * T + q] = st.m[qt] * LN2F + __logf(lsum);
            }
        }
        {   // the next head's queries: their round trip lies under the barrier
            const unsigned short* qbase2 = a.q + (long)m2 * T * a.q_ld + h2 * ATT_D;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                qf[0][s] = ap_frag_global(qbase2, a.q_ld, q0, s, lane2, T);
                qf[1][s] = ap_frag_global(qbase2, a.q_ld, q0 + 16, s, lane2, T);
                qf1[0][s] = ap_frag_global(qbase2, a.q_ld, qL0, s, lane2, T);
            }
        }
        AP_STAMP(5);
        __syncthreads();        // this head's slices released, the next head's visible, the odd tile's partials complete
        AP_STAMP(6);
        if (gm.left && tid2 < nvl * 8) {      // combine: thread = (row, 8 columns); the 8 partials in wave order
            const int rr = tid2 >> 3, c = tid2 & 7;
            float M = -1e30f;
            for (int ww = 0; ww < AP_WAVES; ++ww) M = fmaxf(M, mypart[(ww * nvl + rr) * AP_FPART]);
            float L = 0.f, acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int ww = 0; ww < AP_WAVES; ++ww) {
                const float* p = mypart + (ww * nvl + rr) * AP_FPART;
                const float f = __builtin_amdgcn_exp2f(p[0] - M);
                L += p[1] * f;
                const f32x4_t x = *reinterpret_cast<const f32x4_t*>(p + 4 + c * 8), y = *reinterpret_cast<const f32x4_t*>(p + 8 + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] += x[e] * f;
                    acc[4 + e] += y[e] * f;
                }
            }
            const float inv = 1.0f / L;
            const u32x4_t v = {pack_bf16x2(acc[0] * inv, acc[1] * inv), pack_bf16x2(acc[2] * inv, acc[3] * inv),
                               pack_bf16x2(acc[4] * inv, acc[5] * inv), pack_bf16x2(acc[6] * inv, acc[7] * inv)};
            *reinterpret_cast<u32x4_t*>(a.out + ((long)m * T + qL0 + rr) * a.out_ld + h * ATT_D + c * 8) = v;
            if (c == 0) a.lse[((long)m * H + h) * T + qL0 + rr] = M * LN2F + __logf(L);
        }
    }
}
static int ap_fwd_lds(int TP, int nvl) { return 4 * TP * 128 + 2 * AP_WAVES * nvl * AP_FPART * 4; }


int launch_attn_fwd_pair(const AttnArgs& a, int batch, hipStream_t st) {      // returns 1 when the shape is not taken
    ApGeom gm;
    int TP;
    if (a.Tq != a.Tk || a.causal || a.klen || a.rel_emb || a.drop_p > 0.f || !ap_geom(a.Tq, &gm, &TP)) return 1;
    if ((a.q_ld | a.kv_ld | a.out_ld) % 8) return 1;
    if ((long)a.Tq * a.q_ld * 2 >= (1L << 31) || (long)a.Tq * a.kv_ld * 2 >= (1L << 31)) return 1;
    gm.total_heads = batch * a.H;
    static int ncu = 0;
    if (!ncu) {
        int d = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&prop, d) != hipSuccess) return UNIIR_ELAUNCH;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
#ifdef UNIIR_EXP_BUILD
    if (g_att_exp & 64) return 1;
#endif
    const int sm = ap_fwd_lds(TP, gm.nvl);
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_pair_kernel<288>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_fwd_pair_kernel<224>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const int grid = gm.total_heads < ncu ? gm.total_heads : ncu;
    if (TP == 288) hipLaunchKernelGGL(attn_fwd_pair_kernel<288>, dim3(grid), dim3(AP_THREADS), sm, st, a, gm);
    else hipLaunchKernelGGL(attn_fwd_pair_kernel<224>, dim3(grid), dim3(AP_THREADS), sm, st, a, gm);
    HIP_LAUNCH_CHECK();
    return UNIIR_OK;
}

