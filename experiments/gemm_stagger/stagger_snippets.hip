// removed from csrc/gemm.hip after the round-5 measurement (README.md here)
// --- in gemm_glds_kernel, before the accumulators are zeroed:
    if (p.stag_ticks && (int)blockIdx.x < p.stag_first) {
        // One workgroup per CU and equal tiles: without this every CU reaches its epilogue at the same moment, the epilogues' HBM
        // traffic comes in bursts with the matrix pipes idle, and the main loops run with HBM idle.  Phase groups interleave them.
        const unsigned ph = (blockIdx.x >> 3) % (unsigned)p.stag_p;          // blockIdx & 7 is the XCD: alternate inside an XCD
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz, independent of the shader clock
        const unsigned long long d = (unsigned long long)ph * p.stag_ticks;
        while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(32);
    }

// --- host side:
// Tuning knobs of the 256x256 kernel's start stagger (experiments / bench sweeps; defaults below are what ships)
static struct {
    int phases = 0;          // 0 / 1: off
    int ns_kstep = 1450;     // estimated main-loop time per 64-wide K step
    int ns_epi = 6000;       // estimated epilogue time of the tile
    int min_rounds = 6;      // only when a CU runs at least this many tiles (the stagger costs (phases - 1) / phases of a tile at the end)
} g_tune;
extern "C" int uniir_gemm_tune(int32_t key, int32_t value) {
    switch (key) {
        case 0: g_tune.phases = value; break;
        case 1: g_tune.ns_kstep = value; break;
        case 2: g_tune.ns_epi = value; break;
        case 3: g_tune.min_rounds = value; break;
        default: return UNIIR_EINVAL;
    }
    return UNIIR_OK;
}
static int device_cus() {
    static int cus[64] = {};
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 63;
    if (!cus[d]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
        cus[d] = v;
    }
    return cus[d];
}


// --- in gemm_impl:
    a.stag_first = 0; a.stag_p = 1; a.stag_ticks = 0;
    if (g_tune.phases > 1 && d->k_splits == 1 && gemm_shape(a, d->a_tmaj, d->b_tmaj) == 1) {
        const int cus = device_cus();
        const long tiles = (long)((d->M + 255) / 256) * ((d->N + 255) / 256);
        if (tiles >= (long)g_tune.min_rounds * cus) {
            const long period_ns = (long)(d->K / 64) * g_tune.ns_kstep + g_tune.ns_epi;
            a.stag_first = cus;
            a.stag_p = g_tune.phases;
            a.stag_ticks = (unsigned)(period_ns / g_tune.phases / 10);
        }
    }
