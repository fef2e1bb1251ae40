// C ABI of libdvbs2hip.so (include/dvbs2hip.h): handle, tables, workspaces, stream plumbing.
// The arithmetic is in k_ldpc.hip / k_bch.hip / k_front.hip / k_fir.hip.  No CPU fallback.
#include "dvbs2hip_internal.h"
#include "dvbs2_tables_gen.h"
#include <dlfcn.h>
#include <unistd.h>
#include <fcntl.h>
#include <time.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <thread>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>

using namespace dvbs2;

typedef struct { char internal[128]; } dvbs2hip_nccl_id;        // = ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct dvbs2hip_handle {
    // configuration
    int N_ldpc = 0, K_ldpc = 0, K_bch = 0, bps = 0, itl_cols = 1, itl_order = 0;
    int n_sym = 0, pl_frame = 0, max_frames = 1, device = 0;
    int n_ite = 50, early_stop = 1, implem = 0;
    float alpha = 1.f, code_rate = 0.f;
    int fir_T = 0, fir_osf = 1;
    // device state
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool capturing = false;
    bool lat_lds_ok = false;      // the device's LDS holds k_ldpc_lat.hip's image + state (gfx950: 160 KB)
    bool comm_dead = false;       // dvbs2hip_monitor_reduce timed out: the communicator was aborted
    int red_timeout_ms = 0;
    std::vector<hipGraphExec_t> graphs;      // dvbs2hip_graph_end: captured call sequences (a freed slot is nullptr)
    int n_cus = 256;
    LdpcPlan ldpc;
    BchPlan bch;
    float *d_cstl = nullptr;
    uint8_t *d_pl_seq = nullptr;
    float *d_taps_rev = nullptr;
    uint16_t *d_fir_afrag = nullptr, *d_upfir_afrag = nullptr;       // Toeplitz fragments of the split taps for the matrix-core FIR (T <= 81)
    float *d_hist[2] = {nullptr, nullptr};
    void *d_hist_all = nullptr;      // one allocation behind d_hist[] and d_uphist[]
    size_t hist_stride = 0;
    int hist_cur = 0;
    float *d_taps = nullptr;            // natural order (shaping filter)
    float *d_uphist[2] = {nullptr, nullptr};
    int uphist_cur = 0;
    unsigned long long *d_ctr = nullptr;
    float *d_gwork = nullptr;
    // TX mirror (N1)
    uint32_t *d_enc_tab = nullptr;
    int32_t *d_enc_deg = nullptr;
    float *d_plh = nullptr;
    int enc_stride = 0;
    unsigned long long *d_bch_tab = nullptr, *d_bch_shift = nullptr;
    uint32_t *d_syn_pos = nullptr, *d_prbs_s = nullptr;     // (round 5) the LDPC kernel's BCH verification in the fused chain: row records + reduction table (LdpcKParams::syn_tab), PRBS words by (storage row, wave)
    int syn_words = 0, syn_rows = 0;
    std::map<int, DevBuf> bufs;        // lazily grown staging / intermediate buffers
    // frame synchronizer (N4): device-resident state of Synchronizer_frame_DVBS2_fast
    struct {
        bool ready = false;
        float alpha = 0.9f, trigger = 30.f;    // factory defaults, Factory/Module/Synchronizer_frame/Synchronizer_frame.hpp:26-27
        int vec_width = 8;                     // mipp::N<float>() of the reference build (AVX2)
        int nbuff2 = 0;
        float *xh[2] = {nullptr, nullptr};     // last 64 input samples (the correlators' memories + reg_channel)
        float *sofh[2] = {nullptr, nullptr};   // last 64 cor_SOF samples (SOF_PLSC_delay)
        float *cv = nullptr;                   // corr_vec
        float *buff2[2] = {nullptr, nullptr};  // output_delay.buff2
        int *st[2] = {nullptr, nullptr};       // output_delay {head2, first_time}
        float *yprev[2] = {nullptr, nullptr};  // the last output frame of the previous call
        unsigned long long *keys = nullptr;    // arg max keys, max_frames x ceil(pl_frame / 64)
        float *metric = nullptr;               // max_corr of the last frame
        uint16_t *frag = nullptr;              // band fragments of the two correlators for the matrix cores (k_sync_mfma.hip)
        int xh_cur = 0, sofh_cur = 0, od_cur = 0, yp_cur = 0;
    } sfm;
    // L&R fine frequency synchronizer (N4): damped autocorrelation R_l, alpha (factory default 0.999)
    // host sockets the integrator has pinned (dvbs2hip_host_register): base address -> bytes; copy streams + events of
    // the chunked host-form pipeline (H2D of chunk i+1 | kernels of chunk i | D2H of chunk i-1)
    std::map<uintptr_t, size_t> pinned;
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> ev_pipe;
    int ldpc_sched = DVBS2HIP_SCHED_QC;
    int sep = 0, sep_ax[2] = {0, 0};       // separable 2-bit constellation: linear exact LLRs (k_front.hip, demap_sep2)
    float sep_g[2] = {0.f, 0.f}, sep_h[2] = {0.f, 0.f};
    int fir_kernel = DVBS2HIP_FIR_AUTO;
    float *d_nat_work = nullptr;       // natural-order LDPC: frame-interleaved image + state, ceil(max_frames / 64) groups
    float *d_lr_R = nullptr;
    float lr_alpha = 0.999f;
    int lr_timeouts = 0;               // launches whose rotation had to be repeated (dvbs2hip_sync_lr_timeouts)
    // L&R, recurrence and rotation in one launch (k_sync.hip, sff_lr_fused_kernel): the error word a rotating workgroup sets when it gives up waiting for the
    // recurrence -- host-mapped memory, so the host reads it at its synchronisation points without a copy
    // (round 5) every outstanding L&R call has its OWN error word and its OWN estimate buffer (LR_SLOTS in rotation): a later L&R or pilot-phase call
    // neither consumes an earlier call's error word nor overwrites the estimates its repair needs
    static constexpr int LR_SLOTS = 4;
    uint32_t *lr_err_host = nullptr, *lr_err_dev = nullptr;          // [LR_SLOTS]
    struct { const float *x = nullptr; float *y = nullptr; int n = 0, F = 0; bool pending = false; } lr_slot[LR_SLOTS];      // device-form calls not yet looked at: what dvbs2hip_synchronize re-rotates after a timeout
    int lr_next = 0;
    float *d_hist_zero = nullptr, *d_hist_junk = nullptr;     // filter2: zero history in, discarded history out
    // monitor reduction over RCCL (one process per GPU): communicator + the 3 x uint64 receive buffer
    void *nccl_comm = nullptr;
    unsigned long long *d_red = nullptr;
    unsigned long long *h_red = nullptr;      // pinned: the reduced counters' way back (a copy into pageable memory would wait for the stream itself -- and for a dead peer for ever)
    int red_rank = 0, red_world = 1;
    // timing
    float nco_nu = 0.f, nco_omega = 0.f;                   // Synchronizer_freq_coarse in the transmission phase: Multiplier_sine_ccc_naive's nu / omega and its sample counter n
    uint32_t nco_n = 0;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[DVBS2HIP_K_COUNT];
    std::string err;
    std::string ldpc_name;
};

namespace {

int fail(dvbs2hip_t *h, int code, const std::string &msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(h, expr)                                                                       \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(h, DVBS2HIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

enum BufId { B_IN = 0, B_OUT, B_AUX0, B_AUX1, B_AUX2, B_AUX3, B_LLR, B_PACKED, B_EST, B_CWD0, B_CWD1, B_INFO, B_SIG, B_TXBCH, B_TXLDPC,
             B_SFM_CORR, B_SFM_MET, B_SFM_SOF, B_SFM_PLSC, B_SFM_DLY, B_SFM_DTAB, B_SFF_TMP, B_SFF_OUT, B_FLT2, B_MON_BE, B_MON_OUT, B_BCHFLAG, B_ORDER, B_LR_TMP0, B_LR_TMP1, B_LR_TMP2, B_LR_TMP3, B_SFM_SCR, B_SFM_NEED };

int ensure(dvbs2hip_t *h, int id, size_t bytes, void **out)
{
    DevBuf &b = h->bufs[id];
    if (b.bytes < bytes) {
        if (b.p) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
        hipError_t e = hipMalloc(&b.p, bytes);
        if (e != hipSuccess) return fail(h, DVBS2HIP_ENOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
        b.bytes = bytes;
    }
    *out = b.p;
    return 0;
}

// every entry point that touches HIP selects the handle's device first: a process may drive several GPUs, one handle each
int enter(dvbs2hip_t *h)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, DVBS2HIP_EHIP, "hipSetDevice failed");
    return 0;
}

int check_frames(dvbs2hip_t *h, int n_frames)
{
    int r0 = enter(h); if (r0) return r0;
    if (n_frames < 1 || n_frames > h->max_frames)
        return fail(h, DVBS2HIP_EINVAL, "'n_frames' has to be in [1, max_frames] ('n_frames' = " + std::to_string(n_frames) +
                                            ", 'max_frames' = " + std::to_string(h->max_frames) + ").");
    return 0;
}

// ---- host-socket pipeline for PINNED sockets: the batch goes through in chunks, copies on two copy streams (the two DMA
// directions run together), kernels on the handle's stream, so a call costs about max(H2D, D2H) instead of H2D + kernels + D2H
struct HostCopy { const void *src; void *dst; size_t bytes_per_frame; };       // H2D: src = host, dst = device; D2H: src = device, dst = host

static bool host_is_pinned(const dvbs2hip_t *h, const void *p, size_t bytes)
{
    if (!p || h->pinned.empty()) return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    auto it = h->pinned.upper_bound(a);
    if (it == h->pinned.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
}

template <typename Fn>
static int host_pipeline(dvbs2hip_t *h, int F, const std::vector<HostCopy> &ins, const std::vector<HostCopy> &outs, Fn dev_call)
{
    const int n_chunks = F >= 1024 ? 8 : F >= 128 ? 4 : 1;
    const int chunk = (F + n_chunks - 1) / n_chunks;
    if (!h->s_in) { HIPCHK(h, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking)); HIPCHK(h, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking)); }
    while ((int)h->ev_pipe.size() < 2 * n_chunks) { hipEvent_t e; HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ev_pipe.push_back(e); }
    HIPCHK(h, hipStreamSynchronize(h->stream));             // what the handle's stream still does with the staging buffers
    int c = 0;
    for (int f0 = 0; f0 < F; f0 += chunk, c++) {
        const int nf = F - f0 < chunk ? F - f0 : chunk;
        for (const HostCopy &x : ins)
            HIPCHK(h, hipMemcpyAsync((char *)x.dst + (size_t)f0 * x.bytes_per_frame, (const char *)x.src + (size_t)f0 * x.bytes_per_frame,
                                     (size_t)nf * x.bytes_per_frame, hipMemcpyHostToDevice, h->s_in));
        HIPCHK(h, hipEventRecord(h->ev_pipe[2 * c], h->s_in));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_pipe[2 * c], 0));
        int r = dev_call(f0, nf);
        if (r) { (void)hipStreamSynchronize(h->s_in); (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->s_out); return r; }
        HIPCHK(h, hipEventRecord(h->ev_pipe[2 * c + 1], h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->s_out, h->ev_pipe[2 * c + 1], 0));
        for (const HostCopy &x : outs)
            if (x.dst)
                HIPCHK(h, hipMemcpyAsync((char *)x.dst + (size_t)f0 * x.bytes_per_frame, (const char *)x.src + (size_t)f0 * x.bytes_per_frame,
                                         (size_t)nf * x.bytes_per_frame, hipMemcpyDeviceToHost, h->s_out));
    }
    HIPCHK(h, hipStreamSynchronize(h->s_out));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}


struct Timer {      // RAII: events around one kernel launch when timing is on
    dvbs2hip_t *h; int k; hipEvent_t a = nullptr, b = nullptr;
    Timer(dvbs2hip_t *h_, int k_) : h(h_), k(k_)
    {
        if (!h->timing) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, h->stream);
    }
    ~Timer()
    {
        if (!a) return;
        (void)hipEventRecord(b, h->stream);
        h->ev[k].push_back({a, b});
    }
};

void pl_sequence(std::vector<uint8_t> &seq)
{
    // ETSI EN 302 307 5.5.4, n = 0; equals PL_RAND_SEQ (Scrambler_PL.hpp:54-4207)
    const int P = (1 << 18) - 1;
    std::vector<uint8_t> x(P), y(P);
    for (int i = 0; i < 18; i++) { x[i] = i == 0; y[i] = 1; }
    for (int i = 0; i + 18 < P; i++) { x[i + 18] = x[i + 7] ^ x[i]; y[i + 18] = y[i + 10] ^ y[i + 7] ^ y[i + 5] ^ y[i]; }
    seq.resize(66420);
    for (int i = 0; i < 66420; i++) {
        const int i2 = (i + 131072) % P;
        seq[i] = (uint8_t)(2 * (x[i2] ^ y[i2]) + (x[i] ^ y[i]));
    }
}

void bb_prbs(int K, std::vector<uint32_t> &out)
{
    // Scrambler_BB.hpp:28 init, Scrambler_BB.hxx:56-64 step
    int l[15] = {1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0};
    out.assign((K + 31) / 32, 0u);
    for (int i = 0; i < K; i++) {
        const int fb = l[14] ^ l[13];
        for (int j = 14; j > 0; j--) l[j] = l[j - 1];
        l[0] = fb;
        if (fb) out[i >> 5] |= 1u << (i & 31);
    }
}

template <typename T>
int upload(dvbs2hip_t *h, T **dst, const T *src, size_t n)
{
    HIPCHK(h, hipMalloc((void **)dst, n * sizeof(T)));
    HIPCHK(h, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

extern "C" {

int dvbs2hip_cfg_from_modcod(const char *modcod, dvbs2hip_cfg *cfg)
{
    if (!cfg) return DVBS2HIP_EINVAL;
    std::string name = modcod ? modcod : "";
    if (name.empty()) name = "QPSK-S_8/9";
    for (int i = 0; i < DVBS2_N_MODCODS; i++) {
        const dvbs2_modcod_row &r = dvbs2_modcod_rows[i];
        if (name != r.name) continue;
        memset(cfg, 0, sizeof *cfg);
        cfg->N_ldpc = r.N_ldpc; cfg->K_ldpc = r.K_ldpc; cfg->K_bch = r.K_bch;
        cfg->ldpc_n_rows = r.ldpc_n_rows; cfg->ldpc_row_ptr = r.rp; cfg->ldpc_addr = r.ad;
        cfg->ldpc_n_ite = 50; cfg->ldpc_implem = DVBS2HIP_IMPLEM_SPA; cfg->ldpc_alpha = 1.0f; cfg->ldpc_early_stop = 1;
        cfg->bch_m = r.bch_m; cfg->bch_t = r.bch_t; cfg->bch_prim = r.prim;
        cfg->bps = r.bps; cfg->cstl = r.cstl;
        cfg->itl_cols = r.itl_cols; cfg->itl_order = r.itl_order;
        cfg->fir_n_taps = 81; cfg->fir_taps = rrc_taps_81; cfg->fir_osf = 2;
        cfg->max_frames = 1; cfg->device = 0; cfg->stream = nullptr; cfg->ldpc_lds_groups = -1;
        for (int k = 0; k < 7; k++) cfg->pls[k] = r.pls[k];
        return DVBS2HIP_OK;
    }
    g_create_error = name + " mod-cod scheme not supported.";
    return DVBS2HIP_EINVAL;
}

int dvbs2hip_device_count(int32_t *count)
{
    if (!count) return DVBS2HIP_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return 0;
}

int dvbs2hip_create(const dvbs2hip_cfg *cfg, dvbs2hip_t **out)
{
    if (!cfg || !out) return fail(nullptr, DVBS2HIP_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->max_frames < 1) return fail(nullptr, DVBS2HIP_EINVAL, "'max_frames' has to be greater than 0");
    if (cfg->max_frames > 65534) return fail(nullptr, DVBS2HIP_EINVAL, "'max_frames' has to be at most 65534 (frames are the second grid dimension of several kernels)");
    if (cfg->bps < 1 || cfg->bps > 5 || !cfg->cstl) return fail(nullptr, DVBS2HIP_EINVAL, "'bps' has to be in [1,5] with a constellation");
    if (cfg->N_ldpc <= 0 || cfg->N_ldpc % cfg->bps) return fail(nullptr, DVBS2HIP_EINVAL, "'N_ldpc' has to be a positive multiple of 'bps'");
    if (cfg->itl_cols > 1 && cfg->N_ldpc % cfg->itl_cols) return fail(nullptr, DVBS2HIP_EINVAL, "'N_ldpc' has to be a multiple of 'itl_cols'");
    if (cfg->ldpc_implem != DVBS2HIP_IMPLEM_NMS && cfg->ldpc_implem != DVBS2HIP_IMPLEM_MS && cfg->ldpc_implem != DVBS2HIP_IMPLEM_SPA && cfg->ldpc_implem != DVBS2HIP_IMPLEM_SPA_TANH && cfg->ldpc_implem != DVBS2HIP_IMPLEM_SPA_EXACT)
        return fail(nullptr, DVBS2HIP_EUNSUPPORTED, "LDPC implem not supported (NMS, MS, SPA, SPA_TANH and SPA_EXACT only)");
    if (!cfg->ldpc_row_ptr || !cfg->ldpc_addr || !cfg->bch_prim) return fail(nullptr, DVBS2HIP_EINVAL, "missing code tables");
    if (cfg->ldpc_n_ite < 1) return fail(nullptr, DVBS2HIP_EINVAL, "'ldpc_n_ite' has to be greater than 0");
    if (cfg->fir_n_taps < 0 || cfg->fir_n_taps > 257 || (cfg->fir_n_taps > 0 && !cfg->fir_taps))
        return fail(nullptr, DVBS2HIP_EINVAL, "'fir_n_taps' has to be in [0,257] with taps");

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1)
        return fail(nullptr, DVBS2HIP_ENODEVICE, "no HIP device available (libdvbs2hip has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= n_dev) return fail(nullptr, DVBS2HIP_EINVAL, "'device' out of range");

    dvbs2hip_t *h = new (std::nothrow) dvbs2hip_handle;
    if (!h) return fail(nullptr, DVBS2HIP_ENOMEM, "out of host memory");
#define CREATE_FAIL(code, msg) do { std::string m__ = (msg); dvbs2hip_destroy(h); return fail(nullptr, code, m__); } while (0)
#define CREATE_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) CREATE_FAIL(DVBS2HIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } while (0)
    h->device = cfg->device;
    CREATE_HIP(hipSetDevice(h->device));
    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, h->device));
    h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; h->own_stream = false; }
    else { CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }

    h->N_ldpc = cfg->N_ldpc; h->K_ldpc = cfg->K_ldpc; h->K_bch = cfg->K_bch; h->bps = cfg->bps;
    h->itl_cols = cfg->itl_cols; h->itl_order = cfg->itl_order; h->max_frames = cfg->max_frames;
    h->n_sym = cfg->N_ldpc / cfg->bps;
    h->pl_frame = 90 * (h->n_sym / 90 + 1) + (h->n_sym / (16 * 90)) * 36;      // DVBS2.cpp:351-355
    h->n_ite = cfg->ldpc_n_ite; h->early_stop = cfg->ldpc_early_stop ? 1 : 0; h->implem = cfg->ldpc_implem;
    h->alpha = cfg->ldpc_implem == DVBS2HIP_IMPLEM_MS ? 1.0f : cfg->ldpc_alpha;
    h->code_rate = (float)cfg->K_bch / (float)cfg->N_ldpc;                      // TX_RX_BB/main.cpp:142
    if (h->n_sym % 90) CREATE_FAIL(DVBS2HIP_EINVAL, "'N_ldpc / bps' has to be a multiple of the 90-symbol slot");

    // ---- LDPC
    // gfx950: a workgroup may own the CU's whole 160 KiB LDS (MI355X_MICROARCH "LDS")
    size_t lds_limit = strstr(prop.gcnArchName, "gfx950") ? 160 * 1024 : prop.sharedMemPerBlock;
    if (const char *ev = getenv("DVBS2HIP_LDS_LIMIT")) lds_limit = (size_t)atol(ev);
    if (lds_limit < 32 * 1024) lds_limit = 32 * 1024;
    h->lat_lds_ok = lds_limit >= 160 * 1024;
    lds_limit -= 512;                                    // static LDS of the kernel + slack
    std::string e = ldpc_build_plan(h->ldpc, cfg->N_ldpc, cfg->K_ldpc, cfg->ldpc_n_rows, cfg->ldpc_row_ptr, cfg->ldpc_addr,
                                    cfg->ldpc_lds_groups, lds_limit, cfg->ldpc_implem == DVBS2HIP_IMPLEM_SPA ? 3 : cfg->ldpc_implem == DVBS2HIP_IMPLEM_SPA_TANH ? 2 : cfg->ldpc_implem == DVBS2HIP_IMPLEM_SPA_EXACT ? 1 : 0, cfg->max_frames <= h->n_cus);
    if (!e.empty()) CREATE_FAIL(DVBS2HIP_EINVAL, e);
    LdpcPlan &lp = h->ldpc;
    if (upload(h, &lp.d_entries, lp.entries.data(), lp.entries.size()) ||
        upload(h, &lp.d_layer_deg, lp.layer_deg.data(), lp.layer_deg.size()) ||
        upload(h, &lp.d_layer_lvl, lp.layer_lvl.data(), lp.layer_lvl.size()) ||
        upload(h, &lp.d_groups, lp.groups.data(), lp.groups.size()))
        CREATE_FAIL(DVBS2HIP_EHIP, h->err);
    if (lp.fast_wg8) {
        if (upload(h, &lp.d_w8_tab, lp.w8_tab.data(), lp.w8_tab.size()) || upload(h, &lp.d_w8_rows, lp.w8_rows.data(), lp.w8_rows.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
        if (!lp.w8_atab.empty() && upload(h, &lp.d_w8_atab, lp.w8_atab.data(), lp.w8_atab.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
        CREATE_HIP(hipMalloc((void **)&lp.d_cu_ctr, (LDPC_CU_CTR_WORDS + LDPC_PROF_WORDS) * sizeof(uint32_t)));
        CREATE_HIP(hipMemset(lp.d_cu_ctr, 0, LDPC_CU_CTR_WORDS * sizeof(uint32_t)));
    }
    lp.grid_max = ldpc_blocks_per_cu(lp) * h->n_cus;
    lp.n_cus = h->n_cus;
    if (const char *ev = getenv("DVBS2HIP_LDPC_GRID_MAX")) { const int g = atoi(ev); if (g >= 1 && g < lp.grid_max) lp.grid_max = g; }   // scaling experiments
    if (lp.gwork_words > 0) CREATE_HIP(hipMalloc((void **)&h->d_gwork, (size_t)lp.grid_max * lp.gwork_words * sizeof(float)));

    // ---- BCH
    e = bch_build_plan(h->bch, cfg->bch_m, cfg->bch_prim, cfg->bch_t, cfg->K_ldpc, cfg->K_bch);
    if (!e.empty()) CREATE_FAIL(DVBS2HIP_EINVAL, e);
    if (upload(h, &h->bch.d_exp, h->bch.exp_.data(), h->bch.exp_.size()) ||
        upload(h, &h->bch.d_log, h->bch.log_.data(), h->bch.log_.size()) ||
        upload(h, &h->bch.d_syn_tab, h->bch.syn_tab.data(), h->bch.syn_tab.size()))
        CREATE_FAIL(DVBS2HIP_EHIP, h->err);
    std::vector<uint32_t> prbs;
    bb_prbs(cfg->K_bch, prbs);
    {
        const int n_rows = cfg->K_ldpc / 360;
        std::vector<uint32_t> rw((size_t)n_rows * 6 * 2, 0u);
        for (int i = 0; i < cfg->K_bch; i++)
            if ((prbs[i >> 5] >> (i & 31)) & 1u) { const int g = i / 360, e = i % 360, w = e >> 6, l = e & 63; rw[(size_t)(g * 6 + w) * 2 + (l >> 5)] |= 1u << (l & 31); }
        if (upload(h, &h->bch.d_prbs_rw, rw.data(), rw.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
    }
    if (upload(h, &h->bch.d_prbs, prbs.data(), prbs.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);

    // ---- TX mirror tables: encoder layer table, BCH generator, PLHEADER
    {
        const int M = cfg->N_ldpc - cfg->K_ldpc, q = M / 360;
        std::vector<std::vector<uint32_t>> lay(q);
        for (int g = 0; g < cfg->ldpc_n_rows; g++)
            for (int p = cfg->ldpc_row_ptr[g]; p < cfg->ldpc_row_ptr[g + 1]; p++)
                lay[cfg->ldpc_addr[p] % q].push_back((uint32_t)(cfg->ldpc_addr[p] / q) | ((uint32_t)g << 9));
        size_t stride = 1;
        for (auto &l : lay) stride = std::max(stride, l.size());
        std::vector<uint32_t> tab((size_t)q * stride, 0u);
        std::vector<int32_t> deg(q);
        for (int r = 0; r < q; r++) { deg[r] = (int32_t)lay[r].size(); std::copy(lay[r].begin(), lay[r].end(), tab.begin() + (size_t)r * stride); }
        h->enc_stride = (int)stride;
        if (upload(h, &h->d_enc_tab, tab.data(), tab.size()) || upload(h, &h->d_enc_deg, deg.data(), deg.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
        const std::vector<uint8_t> g = bch_generator(h->bch);
        if ((int)g.size() - 1 != cfg->K_ldpc - cfg->K_bch || g.size() > 193) CREATE_FAIL(DVBS2HIP_EINVAL, "BCH generator degree does not match N_bch - K_bch");
        {   // byte-wise encoder table: T[u] = (u(x) x^r) mod g(x), u's bit 7 = highest degree
            const int r = (int)g.size() - 1;
            unsigned long long gl[3] = {0, 0, 0};
            for (int i = 0; i < r; i++) if (g[i]) gl[i / 64] |= 1ull << (i % 64);
            std::vector<unsigned long long> tab(256 * 3, 0ull);
            for (int u = 0; u < 256; u++) {
                unsigned long long s[3] = {0, 0, 0};
                for (int b = 7; b >= 0; b--) {
                    const int top = r - 1;
                    const unsigned fb = (unsigned)((s[top / 64] >> (top % 64)) & 1ull) ^ ((u >> b) & 1u);
                    s[2] = (s[2] << 1) | (s[1] >> 63); s[1] = (s[1] << 1) | (s[0] >> 63); s[0] <<= 1;
                    for (int w = 0; w < 3; w++) {       // keep r bits
                        const int lo = 64 * w;
                        if (r <= lo) s[w] = 0; else if (r < lo + 64) s[w] &= (1ull << (r - lo)) - 1ull;
                    }
                    if (fb) { s[0] ^= gl[0]; s[1] ^= gl[1]; s[2] ^= gl[2]; }
                }
                tab[3 * u] = s[0]; tab[3 * u + 1] = s[1]; tab[3 * u + 2] = s[2];
            }
            if (upload(h, &h->d_bch_tab, tab.data(), tab.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
            // segmented division (tx_bchpar_kernel): segment s of TX_BCH_SEG is followed by after_s bytes; shift[s][b] = x^(b + 8 after_s) mod g
            {
                const int SEG = TX_BCH_SEG, nbytes = h->K_bch / 8, L = (nbytes + SEG - 1) / SEG;
                std::vector<unsigned long long> sh((size_t)SEG * r * 3, 0ull);
                std::vector<long long> base(SEG);
                long long nmax = 0;
                for (int sgm = 0; sgm < SEG; sgm++) {
                    const int b1 = std::min(std::min(sgm * L, nbytes) + L, nbytes);
                    base[sgm] = 8ll * (nbytes - b1); nmax = std::max(nmax, base[sgm] + r);
                }
                unsigned long long v[3] = {1ull, 0ull, 0ull};                       // x^n mod g, n = 0, 1, ..
                for (long long n = 0; n < nmax; n++) {
                    for (int sgm = 0; sgm < SEG; sgm++)
                        if (n >= base[sgm] && n < base[sgm] + r) { unsigned long long *d = &sh[((size_t)sgm * r + (size_t)(n - base[sgm])) * 3]; d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }
                    const int top = r - 1;
                    const bool fb = (v[top / 64] >> (top % 64)) & 1ull;
                    v[2] = (v[2] << 1) | (v[1] >> 63); v[1] = (v[1] << 1) | (v[0] >> 63); v[0] <<= 1;
                    for (int w = 0; w < 3; w++) { const int lo = 64 * w; if (r <= lo) v[w] = 0; else if (r < lo + 64) v[w] &= (1ull << (r - lo)) - 1ull; }
                    if (fb) { v[0] ^= gl[0]; v[1] ^= gl[1]; v[2] ^= gl[2]; }
                }
                if (upload(h, &h->d_bch_shift, sh.data(), sh.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
            }
            // (round 5) tables of the BCH verification inside the LDPC kernel's output phase (k_ldpc_wg8.hip, `syn_tab`): position i = 360 g + t of the BCH word is the
            // coefficient of x^(N_bch - 1 - i) = x^(360 (G - 1 - g)) x^(359 - t) (k_bch.hip).  One 32-byte record per row in the kernel's STORAGE order (LDS rows, global
            // rows, register slots) {1440 g, last-row flag, A_g = x^(360 (G - 1 - g)) mod g(x)}, then [LDPC_SYN_RED][nsw]: x^k mod g(x) for the once-per-frame reduction;
            // 4 (deg g <= 128) or 6 little-endian 32-bit words; and the BB descrambler's bits by (storage row, wave)
            if (h->ldpc.fast_wg8 && !h->ldpc.fast_cu1) {
                const LdpcPlan &lp = h->ldpc;
                const int G = cfg->K_ldpc / 360, nsw = r <= 128 ? 4 : 6;
                const bool parked = lp.fast_mode == 4 || lp.fast_mode == 5;
                const int nrp = parked ? ldpc_park_nr(lp.fast_mode) : 0, q_ = lp.q;
                std::vector<int> order;                       // bit-group of every emitted row (-1: empty register slot)
                for (int l = 0; l < lp.w8_nl_info; l++) order.push_back((int)lp.w8_rows[l]);
                for (int l = 0; l < lp.w8_ng_info; l++) order.push_back((int)lp.w8_rows[lp.w8_nl + l]);
                for (int k = 0; k < nrp; k++) { const uint32_t g = lp.w8_rows[lp.w8_nl + lp.w8_ng + q_ + k]; order.push_back(g == 0xFFFFFFFFu ? -1 : (int)g); }
                {   // every information row exactly once
                    std::vector<int> seen(G, 0);
                    for (int g : order) if (g >= 0) { if (g >= G || seen[g]++) CREATE_FAIL(DVBS2HIP_EINVAL, "internal: LDPC plan emits an information row twice or a parity row"); }
                    for (int g = 0; g < G; g++) if (!seen[g]) CREATE_FAIL(DVBS2HIP_EINVAL, "internal: LDPC plan does not emit every information row");
                }
                const int rows = (int)order.size();
                std::vector<uint32_t> ag((size_t)G * 6, 0u), sp((size_t)rows * 8 + (size_t)LDPC_SYN_RED * nsw, 0u);
                const int nmax = std::max(360 * (G - 1), LDPC_SYN_RED - 1);
                unsigned long long v[3] = {1ull, 0ull, 0ull};                       // x^n mod g, n = 0, 1, ..
                for (int n = 0; n <= nmax; n++) {
                    auto put = [&](uint32_t *d) {
                        d[0] = (uint32_t)v[0]; d[1] = (uint32_t)(v[0] >> 32); d[2] = (uint32_t)v[1]; d[3] = (uint32_t)(v[1] >> 32);
                        if (nsw == 6) { d[4] = (uint32_t)v[2]; d[5] = (uint32_t)(v[2] >> 32); }
                    };
                    if (n % 360 == 0 && n / 360 < G) put(&ag[(size_t)(G - 1 - n / 360) * 6]);
                    if (n < LDPC_SYN_RED) put(&sp[(size_t)rows * 8 + (size_t)n * nsw]);
                    const int top = r - 1;
                    const bool fb = (v[top / 64] >> (top % 64)) & 1ull;
                    v[2] = (v[2] << 1) | (v[1] >> 63); v[1] = (v[1] << 1) | (v[0] >> 63); v[0] <<= 1;
                    for (int w = 0; w < 3; w++) { const int lo = 64 * w; if (r <= lo) v[w] = 0; else if (r < lo + 64) v[w] &= (1ull << (r - lo)) - 1ull; }
                    if (fb) { v[0] ^= gl[0]; v[1] ^= gl[1]; v[2] ^= gl[2]; }
                }
                // the descrambler's bits per (first row of a batch, lane): bit k of entry [ks][t] = PRBS bit of information bit 360 g + t, g the row emitted at ks + k (k < 16)
                std::vector<uint32_t> ps((size_t)rows * LDPC_AT_LANES, 0u);
                for (int k = 0; k < rows; k++) {
                    uint32_t *d = &sp[(size_t)k * 8];
                    const int g = order[k];
                    if (g < 0) { d[0] = 0x7FFFF000u; continue; }
                    d[0] = (uint32_t)g * 1440u; d[1] = g == G - 1 ? 1u : 0u;
                    for (int i = 0; i < 6; i++) d[2 + i] = ag[(size_t)g * 6 + i];
                }
                for (int ks = 0; ks < rows; ks++)
                    for (int k = 0; k < 16 && ks + k < rows; k++) {
                        const int g = order[ks + k];
                        if (g < 0) continue;
                        for (int e = 0; e < 360; e++) {
                            const int i = g * 360 + e;
                            if (i < cfg->K_bch && ((prbs[i >> 5] >> (i & 31)) & 1u)) ps[(size_t)ks * LDPC_AT_LANES + e] |= 1u << k;
                        }
                    }
                if (upload(h, &h->d_syn_pos, sp.data(), sp.size()) || upload(h, &h->d_prbs_s, ps.data(), ps.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
                h->syn_words = nsw; h->syn_rows = rows;
            }
        }
        // PLHEADER = 26 SOF + 64 PLS symbols, pi/2-BPSK (Framer.hxx:97-196)
        static const int G[7][32] = {
            {1,0,0,1,0,0,0,0,1,0,1,0,1,1,0,0,0,0,1,0,1,1,0,1,1,1,0,1,1,1,0,1}, {0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1},
            {0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1}, {0,0,0,0,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,0,1,1,1,1},
            {0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1},
            {1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1}};
        static const int SCR[64] = {0,1,1,1,0,0,0,1,1,0,0,1,1,1,0,1,1,0,0,0,0,0,1,1,1,1,0,0,1,0,0,1,0,1,0,1,0,0,1,1,0,1,0,0,0,0,1,0,0,0,1,0,1,1,0,1,1,1,1,1,1,0,1,0};
        static const int SOF[26] = {0,1,1,0,0,0,1,1,0,1,0,0,1,0,1,1,1,0,1,0,0,0,0,0,1,0};
        std::vector<float> plh(180);
        const float a = (float)(1 / std::sqrt(2.0));
        for (int i = 0; i < 13; i++) {
            const int e = 1 - 2 * SOF[2 * i], o = 1 - 2 * SOF[2 * i + 1];
            plh[4 * i] = a * e; plh[4 * i + 1] = a * e; plh[4 * i + 2] = -1 * a * o; plh[4 * i + 3] = a * o;
        }
        for (int i = 0; i < 32; i++) {
            int c = 0;
            for (int r = 0; r < 7; r++) c = (c + (cfg->pls[r] & 1) * G[r][i]) % 2;
            const int e = 1 - 2 * ((c + SCR[2 * i]) % 2), o = 1 - 2 * (((c == 0 ? 1 : 0) + SCR[2 * i + 1]) % 2);
            float *p = &plh[52 + 4 * i];
            if ((cfg->pls[0] & 1) == 0) { p[0] = a * e; p[1] = a * e; p[2] = -1 * a * o; p[3] = a * o; }
            else { p[0] = -1 * a * e; p[1] = a * e; p[2] = -1 * a * o; p[3] = -1 * a * o; }
        }
        if (upload(h, &h->d_plh, plh.data(), plh.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
    }

    // ---- modem: normalise to unit mean energy in fp32 (tools::Constellation_user)
    const int P = 1 << cfg->bps;
    std::vector<float> cs(2 * P);
    float es = 0.f;
    for (int i = 0; i < P; i++) es += cfg->cstl[2 * i] * cfg->cstl[2 * i] + cfg->cstl[2 * i + 1] * cfg->cstl[2 * i + 1];
    const float sc = sqrtf(es / (float)P);
    if (!(sc > 0.f)) CREATE_FAIL(DVBS2HIP_EINVAL, "constellation has zero energy");
    for (int i = 0; i < 2 * P; i++) cs[i] = cfg->cstl[i] / sc;
    if (upload(h, &h->d_cstl, cs.data(), cs.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
    if (cfg->bps == 2 && !getenv("DVBS2HIP_DEMAP_GENERAL")) {
        // bit b on one axis alone, two levels, the two bits on different axes => the four points are the product set
        int ax[2] = {-1, -1};
        float lv[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        for (int b = 0; b < 2; b++)
            for (int a = 0; a < 2 && ax[b] < 0; a++) {
                float v[2] = {0.f, 0.f}; bool have[2] = {false, false}, ok = true;
                for (int s = 0; s < 4 && ok; s++) {
                    const int bit = (s >> b) & 1; const float c = cs[2 * s + a];
                    if (!have[bit]) { v[bit] = c; have[bit] = true; } else if (fabsf(v[bit] - c) > 1e-6f) ok = false;
                }
                if (ok && fabsf(v[0] - v[1]) > 1e-3f) { ax[b] = a; lv[b][0] = v[0]; lv[b][1] = v[1]; }
            }
        if (ax[0] >= 0 && ax[1] >= 0 && ax[0] != ax[1]) {
            h->sep = 1;
            for (int b = 0; b < 2; b++) { h->sep_ax[b] = ax[b]; h->sep_g[b] = 2.0f * (lv[b][0] - lv[b][1]); h->sep_h[b] = lv[b][1] * lv[b][1] - lv[b][0] * lv[b][0]; }
        }
    }
    std::vector<uint8_t> seq;
    pl_sequence(seq);
    if (h->pl_frame - 90 > (int)seq.size()) CREATE_FAIL(DVBS2HIP_EINVAL, "PL frame longer than the scrambling sequence");
    if (upload(h, &h->d_pl_seq, seq.data(), seq.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);

    // ---- matched filter: taps stored reversed (Filter_FIR_ccr.cpp:26-27)
    h->fir_T = cfg->fir_n_taps; h->fir_osf = cfg->fir_osf > 0 ? cfg->fir_osf : 1;
    if (h->fir_T > 0) {
        std::vector<float> rev(h->fir_T);
        for (int i = 0; i < h->fir_T; i++) rev[i] = cfg->fir_taps[h->fir_T - 1 - i];
        if (upload(h, &h->d_taps_rev, rev.data(), rev.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
        if (h->fir_T <= 81) {
            const std::vector<uint16_t> af = fir_mfma_afrag(rev.data(), h->fir_T);
            if (upload(h, &h->d_fir_afrag, af.data(), af.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
        }
        const size_t hb = sizeof(float) * 2 * (size_t)(h->fir_T > 1 ? h->fir_T - 1 : 1);
        // the four history buffers in ONE allocation, [hist 0 | uphist 0 | hist 1 | uphist 1]: dvbs2hip_filter_reset is then one memset of the first two (it makes them the current
        // ones) instead of four fill kernels -- ~50 us of the 250 us a one-frame call sequence takes (tools/r05_latency_trace.sh)
        h->hist_stride = (hb + 255) / 256 * 256;
        CREATE_HIP(hipMalloc((void **)&h->d_hist_all, 4 * h->hist_stride));
        CREATE_HIP(hipMemset(h->d_hist_all, 0, 4 * h->hist_stride));
        for (int i = 0; i < 2; i++) { h->d_hist[i] = (float *)((char *)h->d_hist_all + (size_t)(2 * i) * h->hist_stride); h->d_uphist[i] = (float *)((char *)h->d_hist_all + (size_t)(2 * i + 1) * h->hist_stride); }
        if (upload(h, &h->d_taps, cfg->fir_taps, (size_t)h->fir_T)) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
        if (h->fir_osf == 2) {
            const std::vector<uint16_t> af2 = upfir_mfma_afrag(cfg->fir_taps, h->fir_T);
            if (!af2.empty() && upload(h, &h->d_upfir_afrag, af2.data(), af2.size())) CREATE_FAIL(DVBS2HIP_EHIP, h->err);
        }
    }
    CREATE_HIP(hipMalloc((void **)&h->d_ctr, 3 * sizeof(unsigned long long)));
    CREATE_HIP(hipMemset(h->d_ctr, 0, 3 * sizeof(unsigned long long)));
    CREATE_HIP(hipDeviceSynchronize());
#undef CREATE_FAIL
#undef CREATE_HIP
    *out = h;
    return DVBS2HIP_OK;
}

void dvbs2hip_destroy(dvbs2hip_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream && !h->comm_dead) (void)hipStreamSynchronize(h->stream);      // (a stream behind an aborted all-reduce may never drain)
    (void)dvbs2hip_monitor_reduce_finalize(h);
    for (hipGraphExec_t g : h->graphs) if (g) (void)hipGraphExecDestroy(g);
    for (auto &kv : h->bufs) if (kv.second.p) (void)hipFree(kv.second.p);
    for (int k = 0; k < DVBS2HIP_K_COUNT; k++)
        for (auto &p : h->ev[k]) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &kv : h->pinned) (void)hipHostUnregister(reinterpret_cast<void *>(kv.first));
    for (hipEvent_t e : h->ev_pipe) (void)hipEventDestroy(e);
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    void *sfm_ptrs[] = {h->sfm.xh[0], h->sfm.xh[1], h->sfm.sofh[0], h->sfm.sofh[1], h->sfm.cv, h->sfm.buff2[0], h->sfm.buff2[1], h->sfm.st[0], h->sfm.st[1],
                        h->sfm.yprev[0], h->sfm.yprev[1], h->sfm.keys, h->sfm.metric, h->sfm.frag, h->d_lr_R, h->d_nat_work, h->ldpc.d_nat_tab, h->ldpc.d_nat_haz, h->d_fir_afrag, h->d_upfir_afrag, h->d_bch_shift, h->d_hist_zero, h->d_hist_junk, h->d_red, h->bch.d_prbs_rw};
    for (void *p : sfm_ptrs) if (p) (void)hipFree(p);
    if (h->lr_err_host) (void)hipHostFree(h->lr_err_host);
    if (h->h_red) (void)hipHostFree(h->h_red);
    void *ptrs[] = {h->ldpc.d_cu_ctr, h->ldpc.d_w8_tab, h->ldpc.d_w8_atab, h->ldpc.d_w8_rows, h->ldpc.d_entries, h->ldpc.d_layer_deg, h->ldpc.d_layer_lvl, h->ldpc.d_groups, h->bch.d_syn_tab, h->bch.d_exp, h->bch.d_log,
                    h->bch.d_prbs, h->d_cstl, h->d_pl_seq, h->d_taps_rev, h->d_hist_all, h->d_ctr, h->d_gwork, h->d_enc_tab, h->d_enc_deg, h->d_plh, h->d_bch_tab, h->d_syn_pos, h->d_prbs_s, h->d_taps};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *dvbs2hip_last_error(const dvbs2hip_t *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dvbs2hip_set_ldpc_schedule(dvbs2hip_t *h, int32_t schedule)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (schedule != DVBS2HIP_SCHED_QC && schedule != DVBS2HIP_SCHED_NATURAL) return fail(h, DVBS2HIP_EINVAL, "unknown LDPC schedule");
    if (schedule == DVBS2HIP_SCHED_NATURAL && !h->ldpc.fast)
        return fail(h, DVBS2HIP_EUNSUPPORTED, "the natural-order schedule is implemented for codes with check degree <= 27");
    h->ldpc_sched = schedule;
    return 0;
}

int dvbs2hip_set_filter_kernel(dvbs2hip_t *h, int32_t kernel)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (kernel != DVBS2HIP_FIR_AUTO && kernel != DVBS2HIP_FIR_VALU && kernel != DVBS2HIP_FIR_MFMA) return fail(h, DVBS2HIP_EINVAL, "unknown filter kernel");
    if (kernel == DVBS2HIP_FIR_MFMA && !h->d_fir_afrag) return fail(h, DVBS2HIP_EUNSUPPORTED, "the matrix-core filter takes at most 81 taps");
    h->fir_kernel = kernel;
    return 0;
}

static bool ldpc_lat_ok(const dvbs2hip_t *h, int F);
const char *dvbs2hip_ldpc_kernel_name(const dvbs2hip_t *h)
{
    if (!h) return "";
    if (h->ldpc_sched == DVBS2HIP_SCHED_NATURAL) {      // one lane per frame from 32768 frames on, a check's edges over 4 / 8 lanes below (k_ldpc_nat.hip: ldpc_nat_launch)
        const std::string d = std::to_string(h->ldpc.fast_deg);
        if (h->ldpc.spa) { const_cast<dvbs2hip_t *>(h)->ldpc_name = "ldpc_nat_spa_kernel<" + d + "," + (h->ldpc.spa_rule == 2 ? "2" : "1") + ">"; return h->ldpc_name.c_str(); }
        const_cast<dvbs2hip_t *>(h)->ldpc_name = "ldpc_nat_kernel<" + d + "> / ldpc_nat_part_kernel<" + d + ",4|8> / ldpc_nat_ck_kernel<" + d + ",8|4,1|2|4> by batch size";
        return h->ldpc_name.c_str();
    }
    {
        const LdpcPlan &pl = h->ldpc;
        char buf[96];
        if (ldpc_lat_ok(h, h->max_frames)) { snprintf(buf, sizeof buf, "ldpc_lat_kernel<%d>", pl.fast_deg); const_cast<dvbs2hip_t *>(h)->ldpc_name = buf; return h->ldpc_name.c_str(); }      // (every call of this handle is a small batch)
        if (!pl.fast) snprintf(buf, sizeof buf, "ldpc_layered_nms_kernel<%d,%s,%s>", pl.ent_stride, pl.hybrid ? "true" : "false", pl.c2v_lds ? "true" : "false");
        else if (pl.fast_cu1 && pl.spa) snprintf(buf, sizeof buf, "ldpc_cu1_kernel<%d,%d>", pl.fast_deg, pl.spa_rule == 3 ? 3 : 1);
        else if (pl.fast_cu1) snprintf(buf, sizeof buf, "ldpc_cu1_kernel<%d>", pl.fast_deg);
        else if (pl.spa) snprintf(buf, sizeof buf, "ldpc_wg8_kernel<%d,%d,%d>", pl.fast_deg, pl.fast_mode, pl.spa_rule);
        else snprintf(buf, sizeof buf, "ldpc_wg8_kernel<%d,%d>", pl.fast_deg, pl.fast_mode);
        const_cast<dvbs2hip_t *>(h)->ldpc_name = buf;
    }
    return h->ldpc_name.c_str();
}

int dvbs2hip_reset(dvbs2hip_t *h)
{
    if (!h) return DVBS2HIP_EINVAL;
    int r = dvbs2hip_filter_reset(h);
    if (r) return r;
    return dvbs2hip_monitor_reset(h);
}

int dvbs2hip_set_ldpc_params(dvbs2hip_t *h, int32_t n_ite, float alpha, int32_t early_stop)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (n_ite < 1) return fail(h, DVBS2HIP_EINVAL, "'n_ite' has to be greater than 0");
    if (!(alpha > 0.f)) return fail(h, DVBS2HIP_EINVAL, "'alpha' has to be greater than 0");
    h->n_ite = n_ite; h->alpha = alpha; h->early_stop = early_stop ? 1 : 0;
    return 0;
}

void *dvbs2hip_get_stream(dvbs2hip_t *h) { return h ? (void *)h->stream : nullptr; }

// L&R timeout (sff_lr_fused_kernel): every estimate has been published by the time the launch is over, so the rotation alone is run again (the stores
// the waiting workgroups dropped) -- with the estimates of THAT launch (the slot's own buffer).  The caller has synchronized the stream.  Returns 0 when there was
// nothing to do or the recovery succeeded.
static int lr_check_recover(dvbs2hip_t *h, int slot)
{
    auto &ls = h->lr_slot[slot];
    ls.pending = false;
    if (!h->lr_err_host || !((volatile uint32_t *)h->lr_err_host)[slot]) return 0;
    ((volatile uint32_t *)h->lr_err_host)[slot] = 0u;
    h->lr_timeouts++;
    auto it = h->bufs.find(B_LR_TMP0 + slot);
    if (!ls.x || !ls.y || it == h->bufs.end() || !it->second.p || it->second.bytes < sizeof(float) * 4 * (size_t)ls.F)
        return fail(h, DVBS2HIP_EHIP, "L&R: a rotating workgroup timed out waiting for the recurrence and the call cannot be repeated (buffers unknown); re-run it with DVBS2HIP_LR=unfused");
    HIPCHK(h, sff_lr_recover(ls.x, ls.y, (float *)it->second.p, ls.n, ls.F, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// every device-form L&R call that has not been looked at yet, oldest first (the stream is synchronized here)
static int lr_check_all(dvbs2hip_t *h)
{
    bool any = false;
    for (int i = 0; i < dvbs2hip_handle::LR_SLOTS; i++) any |= h->lr_slot[i].pending;
    if (!any) return 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int r = 0;
    for (int i = 0; i < dvbs2hip_handle::LR_SLOTS; i++) {
        const int s = (h->lr_next + i) % dvbs2hip_handle::LR_SLOTS;          // lr_next is the oldest slot
        if (h->lr_slot[s].pending) { const int ri = lr_check_recover(h, s); if (ri && !r) r = ri; }
    }
    return r;
}

int dvbs2hip_synchronize(dvbs2hip_t *h)
{
    int r0 = enter(h); if (r0) return r0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return lr_check_all(h);      // (device-form L&R calls: their error words are looked at here)
}

// ------------------------------------------------------------------ call sequences as hipGraphs (small batches: the sequence filter -> extract -> rx_bb is six launches
// and a fill for one workgroup's worth of work; captured once per (F, buffers) and replayed it is ONE submission)
int dvbs2hip_graph_begin(dvbs2hip_t *h)
{
    int r0 = enter(h); if (r0) return r0;
    if (h->capturing) return fail(h, DVBS2HIP_EINVAL, "a capture is already open on this handle");
    if (h->timing) return fail(h, DVBS2HIP_EINVAL, "the per-kernel timers are on: their events cannot be recorded into a graph (dvbs2hip_timing_enable(h, 0) first)");
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    h->capturing = true;
    return 0;
}

int dvbs2hip_graph_end(dvbs2hip_t *h, int32_t *graph)
{
    if (!h || !graph) return DVBS2HIP_EINVAL;
    if (!h->capturing) return fail(h, DVBS2HIP_EINVAL, "no capture is open on this handle");
    h->capturing = false;
    hipGraph_t g = nullptr;
    HIPCHK(h, hipStreamEndCapture(h->stream, &g));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIPCHK(h, e);
    size_t k = 0;
    while (k < h->graphs.size() && h->graphs[k]) k++;
    if (k == h->graphs.size()) h->graphs.push_back(ex); else h->graphs[k] = ex;
    *graph = (int32_t)k;
    return 0;
}

int dvbs2hip_graph_launch(dvbs2hip_t *h, int32_t graph)
{
    int r0 = enter(h); if (r0) return r0;
    if (graph < 0 || (size_t)graph >= h->graphs.size() || !h->graphs[(size_t)graph]) return fail(h, DVBS2HIP_EINVAL, "unknown graph");
    HIPCHK(h, hipGraphLaunch(h->graphs[(size_t)graph], h->stream));
    return 0;
}

int dvbs2hip_graph_destroy(dvbs2hip_t *h, int32_t graph)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (graph < 0 || (size_t)graph >= h->graphs.size() || !h->graphs[(size_t)graph]) return fail(h, DVBS2HIP_EINVAL, "unknown graph");
    (void)hipGraphExecDestroy(h->graphs[(size_t)graph]);
    h->graphs[(size_t)graph] = nullptr;
    return 0;
}

int dvbs2hip_get_sizes(const dvbs2hip_t *h, dvbs2hip_sizes *o)
{
    if (!h || !o) return DVBS2HIP_EINVAL;
    o->N_ldpc = h->N_ldpc; o->K_ldpc = h->K_ldpc; o->K_bch = h->K_bch; o->bps = h->bps;
    o->N_xfec_sym = h->n_sym; o->pl_frame_sym = h->pl_frame; o->ldpc_edges = h->ldpc.E; o->ldpc_q = h->ldpc.q;
    return 0;
}

// ------------------------------------------------------------------ a1
static bool env_is(const char *name, char c) { const char *e = getenv(name); return e && e[0] == c; }
// the LDPC kernel can write the chain's output socket itself (descrambled info bits of a frame the BCH stage leaves alone)
// (round 6) a call of at most one frame per CU on a code whose image and packed state fit the LDS can run on the two-lanes-per-check kernel (k_ldpc_lat.hip) -- OPT-IN,
// DVBS2HIP_LDPC_LAT=1: bit-exact, and measured SLOWER than the lone workgroup of k_ldpc_wg8.hip (one 32APSK-S 3/4 frame: 210 against 177 us; QPSK-S 8/9: 178 against 141):
// splitting a check over two lanes halves a wave's work per layer but not the SIMDs' -- the same vector instructions issue from twice the waves (docs/negative_results.md)
static bool ldpc_lat_ok(const dvbs2hip_t *h, int F)
{
    const char *e = getenv("DVBS2HIP_LDPC_LAT");
    const size_t lds = ldpc_lat_lds_bytes(h->ldpc);
    return e && e[0] == '1' && h->ldpc_sched == DVBS2HIP_SCHED_QC && F <= h->n_cus && lds > 0 && lds <= 160 * 1024 - 512 && h->lat_lds_ok;
}

static bool ldpc_writes_info(const dvbs2hip_t *h) { return h->ldpc.fast_wg8 && h->ldpc_sched == DVBS2HIP_SCHED_QC && !getenv("DVBS2HIP_CHAIN_UNFUSED"); }

// (round 5) ... and check the BCH remainder of what it outputs: the BCH stage then decodes the flagged frames only (DVBS2HIP_CHAIN_SYN=0: round 4's form, the BCH stage
// forms the syndromes of every frame from the packed hard decisions)
static bool ldpc_verifies_bch(const dvbs2hip_t *h) { const char *e = getenv("DVBS2HIP_CHAIN_SYN"); return ldpc_writes_info(h) && h->d_syn_pos && !h->ldpc.fast_cu1 && !(e && e[0] == '0'); }

static int ldpc_dev(dvbs2hip_t *h, const float *Y, int8_t *CWD, int32_t *V, uint32_t *packed, float *post, int32_t *ites, int F, int32_t *info_out = nullptr,
                    uint8_t *bch_flag = nullptr, int8_t *cwd_bch = nullptr)
{
    LdpcKParams p;
    memset(&p, 0, sizeof p);
    p.llr = Y; p.bits = V; p.packed = packed; p.cwd = CWD; p.post = post; p.ites = ites; p.gwork = h->d_gwork;
    p.info_out = info_out; p.info_prbs = h->bch.d_prbs_rw; p.K_info = h->K_bch;
    if (info_out && bch_flag) { p.syn_tab = h->d_syn_pos; p.info_prbs_s = h->d_prbs_s; p.syn_words = h->syn_words; p.syn_rows = h->syn_rows; p.bch_flag = bch_flag; p.cwd_bch = cwd_bch; }
    p.n_frames = F; p.n_ite = h->n_ite; p.early_stop = h->early_stop; p.alpha = h->alpha;
    if (h->ldpc_sched == DVBS2HIP_SCHED_NATURAL) {
        LdpcPlan &pl = h->ldpc;
        if (!h->d_nat_work || !pl.d_nat_tab || !pl.d_nat_haz) {
            const size_t groups = ((size_t)h->max_frames + 63) / 64;
            if (!h->d_nat_work && hipMalloc((void **)&h->d_nat_work, groups * ldpc_nat_group_words(pl) * sizeof(float)) != hipSuccess) {
                h->d_nat_work = nullptr;
                return fail(h, DVBS2HIP_ENOMEM, "natural-order LDPC: workspace of " + std::to_string(groups * ldpc_nat_group_words(pl) * 4) + " bytes does not fit");
            }
            // a failed upload leaves its pointer null (or is freed here), so the next call tries again instead of launching with null tables
            if (!pl.d_nat_tab && upload(h, &pl.d_nat_tab, pl.nat_tab.data(), pl.nat_tab.size())) { if (pl.d_nat_tab) { (void)hipFree(pl.d_nat_tab); pl.d_nat_tab = nullptr; } return DVBS2HIP_EHIP; }
            if (!pl.d_nat_haz && upload(h, &pl.d_nat_haz, pl.nat_haz.data(), pl.nat_haz.size())) { if (pl.d_nat_haz) { (void)hipFree(pl.d_nat_haz); pl.d_nat_haz = nullptr; } return DVBS2HIP_EHIP; }
        }
        Timer tm(h, DVBS2HIP_K_LDPC);
        HIPCHK(h, ldpc_nat_launch(pl, p, h->d_nat_work, h->stream));
        return 0;
    }
    if (ldpc_lat_ok(h, F) && !info_out) {
        Timer tm(h, DVBS2HIP_K_LDPC);
        HIPCHK(h, ldpc_lat_launch(h->ldpc, p, h->stream));
        return 0;
    }
    Timer tm(h, DVBS2HIP_K_LDPC);
    // (round 6, measured NEGATIVE and therefore opt-in: DVBS2HIP_LDPC_ORDER=1) with the stopping rule the work queue can hand out the noisiest frames first
    // (frame_order_launch).  Same-box A/B of the reference's configuration, 2 M frames per point: 8.30 / 13.33 / 18.05 Gb/s at 3.6 / 3.7 / 3.8 dB in index order against
    // 8.16 / 13.09 / 17.98 ordered (three clones: 22.5 -> 21.3 at 3.8 dB; QPSK-N: -4 %): the queue already balances the workgroups, the frames that run to the cap are not
    // what a launch waits for -- the two small kernels and the lost locality cost more than the order gives (docs/negative_results.md, round 6).
    if (h->early_stop && h->ldpc.fast_wg8 && F >= (h->ldpc.fast_cu1 ? 2 : 4) * h->n_cus && env_is("DVBS2HIP_LDPC_ORDER", '1')) {
        void *dord;
        int r = ensure(h, B_ORDER, (size_t)F * 8, &dord);
        if (r) return r;
        HIPCHK(h, frame_order_launch(Y, (float *)dord + F, (uint32_t *)dord, F, h->N_ldpc, h->stream));
        p.order = (const uint32_t *)dord;
    }
    HIPCHK(h, ldpc_launch(h->ldpc, p, h->stream));
    return 0;
}

int dvbs2hip_ldpc_decode_siho_dev(dvbs2hip_t *h, const float *Y_N, int8_t *CWD, int32_t *V_K, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!Y_N || !V_K) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    return ldpc_dev(h, Y_N, CWD, V_K, nullptr, nullptr, nullptr, F);
}

int dvbs2hip_ldpc_decode_siho_post(dvbs2hip_t *h, const float *Y_N, int8_t *CWD, int32_t *V_K, float *post, int32_t *ites, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!Y_N || !V_K) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t nin = (size_t)F * h->N_ldpc * 4, nout = (size_t)F * h->K_ldpc * 4;
    void *din, *dout, *dcwd, *dpost = nullptr, *dit = nullptr;
    if ((r = ensure(h, B_IN, nin, &din)) || (r = ensure(h, B_OUT, nout, &dout)) || (r = ensure(h, B_CWD0, F, &dcwd))) return r;
    if (post && (r = ensure(h, B_AUX0, nin, &dpost))) return r;
    if (ites && (r = ensure(h, B_AUX1, (size_t)F * 4, &dit))) return r;
    if (!post && !ites && host_is_pinned(h, Y_N, nin) && host_is_pinned(h, V_K, nout) && (!CWD || host_is_pinned(h, CWD, (size_t)F))) {
        const size_t N = (size_t)h->N_ldpc, K = (size_t)h->K_ldpc;
        return host_pipeline(h, F, {{Y_N, din, N * 4}}, {{dout, V_K, K * 4}, {dcwd, CWD, 1}}, [&](int f0, int nf) {
            return ldpc_dev(h, (const float *)din + (size_t)f0 * N, (int8_t *)dcwd + f0, (int32_t *)dout + (size_t)f0 * K, nullptr, nullptr, nullptr, nf);
        });
    }
    HIPCHK(h, hipMemcpyAsync(din, Y_N, nin, hipMemcpyHostToDevice, h->stream));
    if ((r = ldpc_dev(h, (const float *)din, (int8_t *)dcwd, (int32_t *)dout, nullptr, (float *)dpost, (int32_t *)dit, F))) return r;
    HIPCHK(h, hipMemcpyAsync(V_K, dout, nout, hipMemcpyDeviceToHost, h->stream));
    if (CWD) HIPCHK(h, hipMemcpyAsync(CWD, dcwd, F, hipMemcpyDeviceToHost, h->stream));
    if (post) HIPCHK(h, hipMemcpyAsync(post, dpost, nin, hipMemcpyDeviceToHost, h->stream));
    if (ites) HIPCHK(h, hipMemcpyAsync(ites, dit, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int dvbs2hip_ldpc_decode_siho(dvbs2hip_t *h, const float *Y_N, int8_t *CWD, int32_t *V_K, int32_t F)
{
    return dvbs2hip_ldpc_decode_siho_post(h, Y_N, CWD, V_K, nullptr, nullptr, F);
}

// ------------------------------------------------------------------ a2
static int bch_dev(dvbs2hip_t *h, const int32_t *Y, const uint32_t *packed, int8_t *CWD, int32_t *V, bool descramble, int F, bool patch_only = false, const uint8_t *flag = nullptr)
{
    BchKParams p;
    memset(&p, 0, sizeof p);
    p.in_bits = Y; p.in_packed = packed; p.out_bits = V; p.cwd = CWD; p.n_frames = F; p.patch_only = patch_only ? 1 : 0; p.flag = flag;
    p.prbs = descramble ? h->bch.d_prbs : nullptr;
    Timer tm(h, DVBS2HIP_K_BCH);
    HIPCHK(h, bch_launch(h->bch, p, h->stream));
    return 0;
}

int dvbs2hip_bch_decode_hiho_dev(dvbs2hip_t *h, const int32_t *Y_N, int8_t *CWD, int32_t *V_K, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!Y_N || !V_K) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    return bch_dev(h, Y_N, nullptr, CWD, V_K, false, F);
}

int dvbs2hip_bch_decode_hiho(dvbs2hip_t *h, const int32_t *Y_N, int8_t *CWD, int32_t *V_K, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!Y_N || !V_K) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t nin = (size_t)F * h->K_ldpc * 4, nout = (size_t)F * h->K_bch * 4;
    void *din, *dout, *dcwd;
    if ((r = ensure(h, B_IN, nin, &din)) || (r = ensure(h, B_OUT, nout, &dout)) || (r = ensure(h, B_CWD0, F, &dcwd))) return r;
    HIPCHK(h, hipMemcpyAsync(din, Y_N, nin, hipMemcpyHostToDevice, h->stream));
    if ((r = bch_dev(h, (const int32_t *)din, nullptr, (int8_t *)dcwd, (int32_t *)dout, false, F))) return r;
    HIPCHK(h, hipMemcpyAsync(V_K, dout, nout, hipMemcpyDeviceToHost, h->stream));
    if (CWD) HIPCHK(h, hipMemcpyAsync(CWD, dcwd, F, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ------------------------------------------------------------------ a3 / a4
static FrontKParams front_params(dvbs2hip_t *h, const float *in, const float *sigma, float *llr, float *est, int F)
{
    FrontKParams p;
    memset(&p, 0, sizeof p);
    p.in = in; p.sigma_in = sigma; p.llr = llr; p.est = est; p.cstl = h->d_cstl; p.pl_seq = h->d_pl_seq;
    p.n_sym = h->n_sym; p.pl_frame = h->pl_frame; p.bps = h->bps; p.itl_cols = h->itl_cols; p.itl_order = h->itl_order;
    p.n_frames = F; p.code_rate = h->code_rate;
    p.sep = h->sep; for (int b = 0; b < 2; b++) { p.sep_ax[b] = h->sep_ax[b]; p.sep_g[b] = h->sep_g[b]; p.sep_h[b] = h->sep_h[b]; }
    return p;
}

static int demod_any_dev(dvbs2hip_t *h, const float *CP, const float *Y1, float *Y2, bool deitl, int F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!CP || !Y1 || !Y2) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    Timer tm(h, DVBS2HIP_K_DEMAP);
    HIPCHK(h, demod_launch(front_params(h, Y1, CP, Y2, nullptr, F), deitl, h->stream));
    return 0;
}

static int demod_any_host(dvbs2hip_t *h, const float *CP, const float *Y1, float *Y2, bool deitl, int F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!CP || !Y1 || !Y2) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t nin = (size_t)F * 2 * h->n_sym * 4, nout = (size_t)F * h->N_ldpc * 4;
    void *din, *dout, *dsig;
    if ((r = ensure(h, B_IN, nin, &din)) || (r = ensure(h, B_OUT, nout, &dout)) || (r = ensure(h, B_SIG, (size_t)F * 4, &dsig))) return r;
    HIPCHK(h, hipMemcpyAsync(din, Y1, nin, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dsig, CP, (size_t)F * 4, hipMemcpyHostToDevice, h->stream));
    if ((r = demod_any_dev(h, (const float *)dsig, (const float *)din, (float *)dout, deitl, F))) return r;
    HIPCHK(h, hipMemcpyAsync(Y2, dout, nout, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int dvbs2hip_demodulate_dev(dvbs2hip_t *h, const float *CP, const float *Y1, float *Y2, int32_t F) { return demod_any_dev(h, CP, Y1, Y2, false, F); }
int dvbs2hip_demodulate(dvbs2hip_t *h, const float *CP, const float *Y1, float *Y2, int32_t F) { return demod_any_host(h, CP, Y1, Y2, false, F); }
int dvbs2hip_demodulate_deinterleave_dev(dvbs2hip_t *h, const float *CP, const float *Y1, float *nat, int32_t F) { return demod_any_dev(h, CP, Y1, nat, true, F); }
int dvbs2hip_demodulate_deinterleave(dvbs2hip_t *h, const float *CP, const float *Y1, float *nat, int32_t F) { return demod_any_host(h, CP, Y1, nat, true, F); }

int dvbs2hip_deinterleave_dev(dvbs2hip_t *h, const float *itl, float *nat, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!itl || !nat) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, deinterleave_launch(itl, nat, h->N_ldpc, h->itl_cols, h->itl_order, F, h->stream));
    return 0;
}

}  // extern "C"

// generic "same-size or two-size elementwise" host wrapper; dev_call(in, out, n_frames).  PER_FRAME: the task treats frames
// independently or as one stream in order, so pinned sockets may go through in overlapped chunks
template <bool PER_FRAME = false, typename Tin, typename Tout, typename Fn>
static int host_wrap(dvbs2hip_t *h, const Tin *in, size_t nin_el, Tout *out, size_t nout_el, int F, Fn dev_call)
{
    int r = check_frames(h, F); if (r) return r;
    if (!in || !out) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t nin = (size_t)F * nin_el * sizeof(Tin), nout = (size_t)F * nout_el * sizeof(Tout);
    void *din, *dout;
    if ((r = ensure(h, B_IN, nin, &din)) || (r = ensure(h, B_OUT, nout, &dout))) return r;
    if (PER_FRAME && host_is_pinned(h, in, nin) && host_is_pinned(h, out, nout))
        return host_pipeline(h, F, {{in, din, nin_el * sizeof(Tin)}}, {{dout, out, nout_el * sizeof(Tout)}},
                             [&](int f0, int nf) { return dev_call((const Tin *)din + (size_t)f0 * nin_el, (Tout *)dout + (size_t)f0 * nout_el, nf); });
    HIPCHK(h, hipMemcpyAsync(din, in, nin, hipMemcpyHostToDevice, h->stream));
    if ((r = dev_call((const Tin *)din, (Tout *)dout, F))) return r;
    HIPCHK(h, hipMemcpyAsync(out, dout, nout, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" {

int dvbs2hip_host_register(dvbs2hip_t *h, void *ptr, size_t bytes)
{
    if (!h || !ptr || !bytes) return DVBS2HIP_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, DVBS2HIP_EHIP, "hipSetDevice failed");
    if (host_is_pinned(h, ptr, bytes)) return 0;
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return fail(h, DVBS2HIP_EHIP, std::string("hipHostRegister: ") + hipGetErrorString(e));
    h->pinned[reinterpret_cast<uintptr_t>(ptr)] = bytes;
    return 0;
}

int dvbs2hip_host_unregister(dvbs2hip_t *h, void *ptr)
{
    if (!h || !ptr) return DVBS2HIP_EINVAL;
    auto it = h->pinned.find(reinterpret_cast<uintptr_t>(ptr));
    if (it == h->pinned.end()) return fail(h, DVBS2HIP_EINVAL, "this address was not registered");
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, DVBS2HIP_EHIP, "hipSetDevice failed");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipHostUnregister(ptr);
    h->pinned.erase(it);
    return 0;
}

int dvbs2hip_deinterleave(dvbs2hip_t *h, const float *itl, float *nat, int32_t F)
{
    return host_wrap<true>(h, itl, h ? h->N_ldpc : 0, nat, h ? h->N_ldpc : 0, F,
                           [&](const float *a, float *b, int nf) { return dvbs2hip_deinterleave_dev(h, a, b, nf); });
}

// ------------------------------------------------------------------ a5
int dvbs2hip_filter_dev(dvbs2hip_t *h, const float *X, float *Y, int32_t n_cplx, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X || !Y) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    if (h->fir_T <= 0) return fail(h, DVBS2HIP_EUNSUPPORTED, "handle was created without filter taps");
    if (n_cplx < 1) return fail(h, DVBS2HIP_EINVAL, "'n_cplx' has to be greater than 0");
    Timer tm(h, DVBS2HIP_K_FIR);
    HIPCHK(h, fir_launch(X, Y, h->d_hist[h->hist_cur], h->d_hist[h->hist_cur ^ 1], h->d_taps_rev, h->fir_kernel == DVBS2HIP_FIR_VALU ? nullptr : h->d_fir_afrag, h->fir_T,
                         (long long)n_cplx * F, h->stream));
    h->hist_cur ^= 1;
    return 0;
}

int dvbs2hip_filter(dvbs2hip_t *h, const float *X, float *Y, int32_t n_cplx, int32_t F)
{
    return host_wrap<true>(h, X, (size_t)2 * (n_cplx > 0 ? n_cplx : 0), Y, (size_t)2 * (n_cplx > 0 ? n_cplx : 0), F,
                           [&](const float *a, float *b, int nf) { return dvbs2hip_filter_dev(h, a, b, n_cplx, nf); });
}

// filter1 / filter2: the reference splits the matched filter over two pipeline stages (Filter_FIR_ccr.cpp:144-294; bound
// RX/main_sched.cpp:199-201): filter1 produces the lower part of every frame and advances the state, filter2 copies Y_N2h and
// produces the upper part from X_N1 alone.  Both are pure functions of their sockets here too (they may sit in different
// pipeline stages, working on different batches at the same time).
int dvbs2hip_filter_split(const dvbs2hip_t *h, int32_t n_cplx)
{
    if (!h || h->fir_T <= 0) return DVBS2HIP_EINVAL;
    const int split = (n_cplx / 2) & ~3;                  // 32-byte aligned rows for the 2-D copies
    return split >= h->fir_T - 1 && split < n_cplx ? split : DVBS2HIP_EINVAL;
}

int dvbs2hip_filter1_dev(dvbs2hip_t *h, const float *X, float *Y, int32_t n_cplx, int32_t F)
{
    if (h && dvbs2hip_filter_split(h, n_cplx) < 0) return fail(h, DVBS2HIP_EINVAL, "filter1 / filter2: half a frame has to hold the filter's memory (n_cplx / 2 >= n_taps - 1)");
    // the lower part is all the reference defines for this socket; the upper part of Y_N2, which the reference leaves as it
    // was, is filled too (one stream pass computes both, and filter2 overwrites it anyway)
    return dvbs2hip_filter_dev(h, X, Y, n_cplx, F);
}

int dvbs2hip_filter2_dev(dvbs2hip_t *h, const float *X, const float *Yh, float *Y, int32_t n_cplx, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X || !Yh || !Y) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    if (h->fir_T <= 0) return fail(h, DVBS2HIP_EUNSUPPORTED, "handle was created without filter taps");
    const int split = dvbs2hip_filter_split(h, n_cplx);
    if (split < 0) return fail(h, DVBS2HIP_EINVAL, "filter1 / filter2: half a frame has to hold the filter's memory (n_cplx / 2 >= n_taps - 1)");
    const size_t hb = sizeof(float) * 2 * (size_t)(h->fir_T > 1 ? h->fir_T - 1 : 1);
    if (!h->d_hist_zero || !h->d_hist_junk) {      // both buffers, zeroed, or neither: the handle only ever sees the complete pair
        float *z = nullptr, *j = nullptr;
        hipError_t e = hipMalloc((void **)&z, hb);
        if (e == hipSuccess) e = hipMalloc((void **)&j, hb);
        if (e == hipSuccess) e = hipMemsetAsync(z, 0, hb, h->stream);
        if (e != hipSuccess) { if (z) (void)hipFree(z); if (j) (void)hipFree(j); HIPCHK(h, e); }
        if (h->d_hist_zero) (void)hipFree(h->d_hist_zero);
        if (h->d_hist_junk) (void)hipFree(h->d_hist_junk);
        h->d_hist_zero = z; h->d_hist_junk = j;
    }
    void *tmp;
    const size_t row = sizeof(float) * 2 * (size_t)n_cplx;
    if ((r = ensure(h, B_FLT2, row * F, &tmp))) return r;
    {   // the upper part of a frame reads nothing before the frame (split >= n_taps - 1): the state is neither used nor advanced
        Timer tm(h, DVBS2HIP_K_FIR);
        HIPCHK(h, fir_launch(X, (float *)tmp, h->d_hist_zero, h->d_hist_junk, h->d_taps_rev, h->fir_kernel == DVBS2HIP_FIR_VALU ? nullptr : h->d_fir_afrag, h->fir_T,
                             (long long)n_cplx * F, h->stream));
    }
    const size_t lo = sizeof(float) * 2 * (size_t)split;
    if (Y != Yh) HIPCHK(h, hipMemcpy2DAsync(Y, row, Yh, row, lo, (size_t)F, hipMemcpyDeviceToDevice, h->stream));       // std::copy(Y_N2h, ..) :224
    HIPCHK(h, hipMemcpy2DAsync((char *)Y + lo, row, (const char *)tmp + lo, row, row - lo, (size_t)F, hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

int dvbs2hip_filter1(dvbs2hip_t *h, const float *X, float *Y, int32_t n_cplx, int32_t F)
{
    if (h && dvbs2hip_filter_split(h, n_cplx) < 0) return fail(h, DVBS2HIP_EINVAL, "filter1 / filter2: half a frame has to hold the filter's memory (n_cplx / 2 >= n_taps - 1)");
    return dvbs2hip_filter(h, X, Y, n_cplx, F);
}

int dvbs2hip_filter2(dvbs2hip_t *h, const float *X, const float *Yh, float *Y, int32_t n_cplx, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X || !Yh || !Y) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const int split = dvbs2hip_filter_split(h, n_cplx);
    if (split < 0) return fail(h, DVBS2HIP_EINVAL, "filter1 / filter2: half a frame has to hold the filter's memory (n_cplx / 2 >= n_taps - 1)");
    const size_t row = sizeof(float) * 2 * (size_t)n_cplx, lo = sizeof(float) * 2 * (size_t)split;
    void *din, *dout;
    if ((r = ensure(h, B_IN, row * F, &din)) || (r = ensure(h, B_OUT, row * F, &dout))) return r;
    HIPCHK(h, hipMemcpyAsync(din, X, row * F, hipMemcpyHostToDevice, h->stream));
    if ((r = dvbs2hip_filter2_dev(h, (const float *)din, (const float *)dout, (float *)dout, n_cplx, F))) return r;     // Yh == Y on the device: lower part untouched
    if (Y != Yh) for (int f = 0; f < F; f++) memcpy((char *)Y + (size_t)f * row, (const char *)Yh + (size_t)f * row, lo);   // std::copy(Y_N2h, ..) :224, lower part
    HIPCHK(h, hipMemcpy2DAsync((char *)Y + lo, row, (const char *)dout + lo, row, row - lo, (size_t)F, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int dvbs2hip_filter_reset(dvbs2hip_t *h)
{
    int r0 = enter(h); if (r0) return r0;
    if (h->fir_T > 1 && h->d_hist_all) {      // buffers 0 become the current ones, zeroed by one memset (they are adjacent); buffers 1 are written before they are read
        HIPCHK(h, hipMemsetAsync(h->d_hist_all, 0, 2 * h->hist_stride, h->stream));
        h->hist_cur = 0; h->uphist_cur = 0;
    }
    return 0;
}

// ------------------------------------------------------------------ N2: shaping filter, channel noise, perfect timing
int dvbs2hip_shape_filter_dev(dvbs2hip_t *h, const float *X, float *Y, int32_t n_cplx, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X || !Y) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    if (h->fir_T <= 0) return fail(h, DVBS2HIP_EUNSUPPORTED, "handle was created without filter taps");
    if (n_cplx < 1) return fail(h, DVBS2HIP_EINVAL, "'n_cplx' has to be greater than 0");
    Timer tm(h, DVBS2HIP_K_FIR);
    HIPCHK(h, upfir_launch(X, Y, h->d_uphist[h->uphist_cur], h->d_uphist[h->uphist_cur ^ 1], h->d_taps, h->fir_kernel == DVBS2HIP_FIR_VALU ? nullptr : h->d_upfir_afrag, h->fir_T, h->fir_osf,
                           (long long)n_cplx * F, h->stream));
    h->uphist_cur ^= 1;
    return 0;
}

int dvbs2hip_shape_filter(dvbs2hip_t *h, const float *X, float *Y, int32_t n_cplx, int32_t F)
{
    const size_t n = (size_t)2 * (n_cplx > 0 ? n_cplx : 0);
    return host_wrap(h, X, n, Y, n * (h ? h->fir_osf : 1), F, [&](const float *a, float *b, int nf) { return dvbs2hip_shape_filter_dev(h, a, b, n_cplx, nf); });
}

int dvbs2hip_add_noise_dev(dvbs2hip_t *h, const float *CP, const float *X, float *Y, uint64_t seed, int32_t n_elmts, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!CP || !X || !Y) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    if (n_elmts < 2 || (n_elmts & 1)) return fail(h, DVBS2HIP_EINVAL, "'n_elmts' has to be a positive even number");
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, awgn_launch(X, Y, CP, seed, n_elmts / 2, F, h->stream));
    return 0;
}

int dvbs2hip_add_noise(dvbs2hip_t *h, const float *CP, const float *X, float *Y, uint64_t seed, int32_t n_elmts, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!CP) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    void *dsig;
    if ((r = ensure(h, B_SIG, (size_t)F * 4, &dsig))) return r;
    HIPCHK(h, hipMemcpyAsync(dsig, CP, (size_t)F * 4, hipMemcpyHostToDevice, h->stream));
    const size_t n = n_elmts > 0 ? (size_t)n_elmts : 0;
    return host_wrap(h, X, n, Y, n, F, [&](const float *a, float *b, int nf) { return dvbs2hip_add_noise_dev(h, (const float *)dsig, a, b, seed, n_elmts, nf); });
}

int dvbs2hip_extract_dev(dvbs2hip_t *h, const float *X, float *Y, int32_t n_cplx_out, int32_t osf, int64_t offset, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X || !Y) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    if (n_cplx_out < 1 || osf < 1) return fail(h, DVBS2HIP_EINVAL, "'n_cplx_out' and 'osf' have to be greater than 0");
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, decimate_launch(X, Y, (long long)n_cplx_out * F, osf, offset, (long long)n_cplx_out * F * osf, h->stream));
    return 0;
}

// ------------------------------------------------------------------ N4: frame synchronizer (Synchronizer_frame_DVBS2_fast)
// DVBS2HIP_SYNC=valu (read at every call): the correlators as fp32 vector sums in the reference's order instead of the matrix cores
static const uint16_t *sfm_frag(dvbs2hip_t *h)
{
    const char *e = getenv("DVBS2HIP_SYNC");
    return e && !strcmp(e, "valu") ? nullptr : h->sfm.frag;
}

static int sfm_state_reset(dvbs2hip_t *h, bool all)
{
    const int n = h->pl_frame;
    auto &S = h->sfm;
    const float one[2] = {1.f, 0.f};                                               // reg_channel = (1, 0), .cpp:19 / :309
    for (int i = 0; i < 2; i++) {
        HIPCHK(h, hipMemsetAsync(S.xh[i], 0, sizeof(float) * 2 * 64, h->stream));
        HIPCHK(h, hipMemsetAsync(S.buff2[i], 0, sizeof(float) * (size_t)S.nbuff2, h->stream));
        const int st[2] = {0, 1};                                                  // head2 = 0, first_time = true
        HIPCHK(h, hipMemcpyAsync(S.st[i], st, sizeof st, hipMemcpyHostToDevice, h->stream));
        if (all) HIPCHK(h, hipMemsetAsync(S.sofh[i], 0, sizeof(float) * 2 * 64, h->stream));
    }
    HIPCHK(h, hipMemcpyAsync(S.xh[S.xh_cur] + 2 * 63, one, sizeof one, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(S.cv, 0, sizeof(float) * (size_t)n, h->stream));
    if (all) { HIPCHK(h, hipMemsetAsync(S.yprev[S.yp_cur], 0, sizeof(float) * 2 * (size_t)n, h->stream)); HIPCHK(h, hipMemsetAsync(S.metric, 0, sizeof(float), h->stream)); }
    HIPCHK(h, hipStreamSynchronize(h->stream));                                    // `one` / `st` live on this stack
    return 0;
}

static int sfm_ready(dvbs2hip_t *h)
{
    auto &S = h->sfm;
    if (S.ready) return 0;
    const int n = h->pl_frame;
    S.nbuff2 = 4 * (n + 1);                                                        // Variable_delay_cc_naive(N, N/2, N/2): buff2(4 (max_delay + 1))
    for (int i = 0; i < 2; i++) {
        HIPCHK(h, hipMalloc((void **)&S.xh[i], sizeof(float) * 2 * 64));
        HIPCHK(h, hipMalloc((void **)&S.sofh[i], sizeof(float) * 2 * 64));
        HIPCHK(h, hipMalloc((void **)&S.buff2[i], sizeof(float) * (size_t)S.nbuff2));
        HIPCHK(h, hipMalloc((void **)&S.st[i], sizeof(int) * 4));
    }
    HIPCHK(h, hipMalloc((void **)&S.cv, sizeof(float) * (size_t)n));
    for (int i = 0; i < 2; i++) HIPCHK(h, hipMalloc((void **)&S.yprev[i], sizeof(float) * 2 * (size_t)n));
    HIPCHK(h, hipMalloc((void **)&S.keys, sizeof(unsigned long long) * (size_t)h->max_frames * (size_t)((n + 63) / 64) + sizeof(float) * 2 * (size_t)((h->max_frames + SYNC_SUB - 1) / SYNC_SUB) * (size_t)n));      // (+ the segments' {A, B} of the average over frames: k_sync.hip)
    HIPCHK(h, hipMalloc((void **)&S.metric, sizeof(float)));
    const std::vector<uint16_t> fr = sync_frag_default();
    HIPCHK(h, hipMalloc((void **)&S.frag, fr.size() * sizeof(uint16_t)));
    HIPCHK(h, hipMemcpy(S.frag, fr.data(), fr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    int r = sfm_state_reset(h, true);
    if (r) return r;
    S.ready = true;
    return 0;
}

int dvbs2hip_sync_frame_set_params(dvbs2hip_t *h, float alpha, float trigger, int32_t vec_width)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (vec_width < 1) return fail(h, DVBS2HIP_EINVAL, "'vec_width' has to be greater than 0");
    h->sfm.alpha = alpha; h->sfm.trigger = trigger; h->sfm.vec_width = vec_width;
    return 0;
}

int dvbs2hip_sync_frame_reset(dvbs2hip_t *h)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, DVBS2HIP_EHIP, "hipSetDevice failed");
    int r = sfm_ready(h); if (r) return r;
    return sfm_state_reset(h, false);                                              // .cpp:304-318: SOF_PLSC_delay keeps its memory
}

int dvbs2hip_sync_frame_synchronize1_dev(dvbs2hip_t *h, const float *X_N1, float *cor_SOF, float *cor_PLSC, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X_N1 || !cor_SOF || !cor_PLSC) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    if ((r = sfm_ready(h))) return r;
    auto &S = h->sfm;
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, sync_corr_launch(X_N1, S.xh[S.xh_cur], S.xh[S.xh_cur ^ 1], sfm_frag(h), cor_SOF, cor_PLSC, (long long)h->pl_frame * F, h->stream));
    S.xh_cur ^= 1;
    return 0;
}

// synchronize2, or (cor_SOF == cor_PLSC == null) the whole one-task synchronize with the correlators fused into the metric
static int sfm_sync2(dvbs2hip_t *h, const float *X_N1, const float *cor_SOF, const float *cor_PLSC, int32_t *DEL, int32_t *FLG, float *TRI, float *Y_N2, int32_t F, const float **SRC = nullptr)
{
    int r = check_frames(h, F); if (r) return r;
    int32_t *delay = DEL;
    const bool fused = !cor_SOF && !cor_PLSC;
    if (!X_N1 || (!fused && (!cor_SOF || !cor_PLSC)) || !delay || (!Y_N2 && !SRC)) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    if ((r = sfm_ready(h))) return r;
    auto &S = h->sfm;
    const int n = h->pl_frame;
    void *corr, *met;
    void *dtab;
    if ((r = ensure(h, B_SFM_CORR, sizeof(float) * (size_t)n * F, &corr)) || (r = ensure(h, B_SFM_MET, sizeof(float) * (size_t)F, &met)) ||
        (r = ensure(h, B_SFM_DTAB, sizeof(int32_t) * (size_t)F, &dtab))) return r;
    Timer tm(h, DVBS2HIP_K_MISC);
    if (TRI) met = TRI;
    const SyncTail tail{S.keys, delay, (float *)met, FLG, S.trigger, (int32_t *)dtab, S.metric, reinterpret_cast<float *>(S.keys + (size_t)h->max_frames * (size_t)((n + 63) / 64))};
    if (fused) {
        HIPCHK(h, sync_corr_metric_launch(X_N1, S.xh[S.xh_cur], S.xh[S.xh_cur ^ 1], sfm_frag(h), S.sofh[S.sofh_cur], S.sofh[S.sofh_cur ^ 1], S.cv, (float *)corr, tail,
                                          n, F, S.alpha, S.vec_width, h->stream));
        S.xh_cur ^= 1;
    } else
        HIPCHK(h, sync_metric_launch(cor_SOF, S.sofh[S.sofh_cur], S.sofh[S.sofh_cur ^ 1], cor_PLSC, S.cv, (float *)corr, tail, n, F, S.alpha, S.vec_width, h->stream));
    S.sofh_cur ^= 1;
    // the delay line is a recurrence from frame to frame made of copies only: resolved per output sample, one launch (k_sync.hip)
    void *need = nullptr;
    if (SRC) {      // located form: only the frames that are not a run of the input stream are materialized (into the handle's scratch); SRC[f] says where frame f starts
        void *scr;
        if ((r = ensure(h, B_SFM_SCR, sizeof(float) * 2 * (size_t)n * F, &scr)) || (r = ensure(h, B_SFM_NEED, sizeof(int32_t) * ((size_t)F + 2), &need))) return r;
        Y_N2 = (float *)scr;
    }
    HIPCHK(h, sync_vdelay_launch(X_N1, S.yprev[S.yp_cur], S.yprev[S.yp_cur ^ 1], Y_N2, S.buff2[S.od_cur], S.buff2[S.od_cur ^ 1], S.st[S.od_cur], S.st[S.od_cur ^ 1],
                                 (const int32_t *)dtab, n, S.nbuff2, F, h->stream, (int32_t *)need, SRC));
    S.od_cur ^= 1;
    S.yp_cur ^= 1;
    return 0;
}

int dvbs2hip_sync_frame_synchronize2_dev(dvbs2hip_t *h, const float *X_N1, const float *cor_SOF, const float *cor_PLSC, int32_t *DEL, int32_t *FLG,
                                         float *TRI, float *Y_N2, int32_t F)
{
    if (h && (!cor_SOF || !cor_PLSC)) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    return sfm_sync2(h, X_N1, cor_SOF, cor_PLSC, DEL, FLG, TRI, Y_N2, F);
}

int dvbs2hip_sync_frame_synchronize_dev(dvbs2hip_t *h, const float *X_N1, int32_t *DEL, int32_t *FLG, float *TRI, float *Y_N2, int32_t F)
{
    // the one-task form: the two correlations are no sockets here and stay on chip (sync_corr_m_kernel); DVBS2HIP_SYNC_UNFUSED keeps
    // the two-task path through device scratch (same numbers)
    if (!getenv("DVBS2HIP_SYNC_UNFUSED")) return sfm_sync2(h, X_N1, nullptr, nullptr, DEL, FLG, TRI, Y_N2, F);
    int r = check_frames(h, F); if (r) return r;
    const size_t nb = sizeof(float) * 2 * (size_t)h->pl_frame * F;
    void *cs, *cp;
    if ((r = ensure(h, B_SFM_SOF, nb, &cs)) || (r = ensure(h, B_SFM_PLSC, nb, &cp))) return r;
    if ((r = dvbs2hip_sync_frame_synchronize1_dev(h, X_N1, (float *)cs, (float *)cp, F))) return r;
    return dvbs2hip_sync_frame_synchronize2_dev(h, X_N1, (const float *)cs, (const float *)cp, DEL, FLG, TRI, Y_N2, F);
}

// (round 5) the frame synchronizer for a consumer of this library: instead of the delayed copy Y_N2 it returns, per frame, WHERE the aligned frame starts -- inside X_N1 for a
// frame that is one run of the input stream (every frame in lock but the first and the last of a call), inside the handle's scratch for the others.  X_N1 and the table stay
// valid until the next synchronizer call on this handle; the frames are 8-byte aligned.  DEL / FLG / TRI and the synchronizer's state are those of `synchronize`.
int dvbs2hip_sync_frame_locate_dev(dvbs2hip_t *h, const float *X_N1, int32_t *DEL, int32_t *FLG, float *TRI, const float **SRC, int32_t F)
{
    if (h && !SRC) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    return sfm_sync2(h, X_N1, nullptr, nullptr, DEL, FLG, TRI, nullptr, F, SRC);
}

int dvbs2hip_sync_frame_synchronize1(dvbs2hip_t *h, const float *X_N1, float *cor_SOF, float *cor_PLSC, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X_N1 || !cor_SOF || !cor_PLSC) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t nb = sizeof(float) * 2 * (size_t)h->pl_frame * F;
    void *din, *cs, *cp;
    if ((r = ensure(h, B_IN, nb, &din)) || (r = ensure(h, B_SFM_SOF, nb, &cs)) || (r = ensure(h, B_SFM_PLSC, nb, &cp))) return r;
    HIPCHK(h, hipMemcpyAsync(din, X_N1, nb, hipMemcpyHostToDevice, h->stream));
    if ((r = dvbs2hip_sync_frame_synchronize1_dev(h, (const float *)din, (float *)cs, (float *)cp, F))) return r;
    HIPCHK(h, hipMemcpyAsync(cor_SOF, cs, nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(cor_PLSC, cp, nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// host-socket forms of synchronize2 / synchronize: DEL, FLG, TRI have one entry per frame (Synchronizer_frame.hxx:42-44); FLG, TRI may be NULL
static int sfm_host(dvbs2hip_t *h, const float *X_N1, const float *cor_SOF, const float *cor_PLSC, int32_t *DEL, int32_t *FLG, float *TRI, float *Y_N2, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X_N1 || !DEL || !Y_N2) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t nb = sizeof(float) * 2 * (size_t)h->pl_frame * F;
    void *din, *dout, *cs = nullptr, *cp = nullptr, *dd;
    if ((r = ensure(h, B_IN, nb, &din)) || (r = ensure(h, B_OUT, nb, &dout)) || (r = ensure(h, B_SFM_DLY, (sizeof(int32_t) * 2 + sizeof(float)) * (size_t)F, &dd))) return r;
    int32_t *d_del = (int32_t *)dd, *d_flg = d_del + F;
    float *d_tri = (float *)(d_flg + F);
    HIPCHK(h, hipMemcpyAsync(din, X_N1, nb, hipMemcpyHostToDevice, h->stream));
    if (cor_SOF) {
        if ((r = ensure(h, B_SFM_SOF, nb, &cs)) || (r = ensure(h, B_SFM_PLSC, nb, &cp))) return r;
        HIPCHK(h, hipMemcpyAsync(cs, cor_SOF, nb, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(cp, cor_PLSC, nb, hipMemcpyHostToDevice, h->stream));
        r = dvbs2hip_sync_frame_synchronize2_dev(h, (const float *)din, (const float *)cs, (const float *)cp, d_del, d_flg, d_tri, (float *)dout, F);
    } else r = dvbs2hip_sync_frame_synchronize_dev(h, (const float *)din, d_del, d_flg, d_tri, (float *)dout, F);
    if (r) return r;
    HIPCHK(h, hipMemcpyAsync(DEL, d_del, sizeof(int32_t) * (size_t)F, hipMemcpyDeviceToHost, h->stream));
    if (FLG) HIPCHK(h, hipMemcpyAsync(FLG, d_flg, sizeof(int32_t) * (size_t)F, hipMemcpyDeviceToHost, h->stream));
    if (TRI) HIPCHK(h, hipMemcpyAsync(TRI, d_tri, sizeof(float) * (size_t)F, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(Y_N2, dout, nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int dvbs2hip_sync_frame_synchronize2(dvbs2hip_t *h, const float *X_N1, const float *cor_SOF, const float *cor_PLSC, int32_t *DEL, int32_t *FLG,
                                     float *TRI, float *Y_N2, int32_t F)
{
    if (h && (!cor_SOF || !cor_PLSC)) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    return sfm_host(h, X_N1, cor_SOF, cor_PLSC, DEL, FLG, TRI, Y_N2, F);
}

int dvbs2hip_sync_frame_synchronize(dvbs2hip_t *h, const float *X_N1, int32_t *DEL, int32_t *FLG, float *TRI, float *Y_N2, int32_t F)
{
    return sfm_host(h, X_N1, nullptr, nullptr, DEL, FLG, TRI, Y_N2, F);
}

// ------------------------------------------------------------------ N4: fine frequency / phase synchronizers (sockets X_N1, FRQ, PHS, Y_N2)
static int sff_call(dvbs2hip_t *h, bool lr, bool host, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X_N1 || !Y_N2) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const int n = h->pl_frame;
    if (n <= 1530) return fail(h, DVBS2HIP_EUNSUPPORTED, "the PL frame holds no pilot block");
    if (lr && !h->d_lr_R) { HIPCHK(h, hipMalloc((void **)&h->d_lr_R, 2 * sizeof(float))); HIPCHK(h, hipMemsetAsync(h->d_lr_R, 0, 2 * sizeof(float), h->stream)); }
    if (lr && !h->lr_err_host) {
        HIPCHK(h, hipHostMalloc((void **)&h->lr_err_host, dvbs2hip_handle::LR_SLOTS * sizeof(uint32_t), hipHostMallocMapped));
        for (int i = 0; i < dvbs2hip_handle::LR_SLOTS; i++) h->lr_err_host[i] = 0u;
        HIPCHK(h, hipHostGetDevicePointer((void **)&h->lr_err_dev, h->lr_err_host, 0));
    }
    const size_t nb = sizeof(float) * 2 * (size_t)n * F;
    void *tmp, *din = nullptr, *dout = nullptr, *dfp = nullptr;
    // the estimates: the pilot-phase synchronizer's buffer is its own; an L&R call takes the next of LR_SLOTS slots (estimates + error word).  A slot whose last
    // device-form call has not been looked at yet (LR_SLOTS such calls without a dvbs2hip_synchronize between them) is looked at first -- one stream synchronization,
    // and only then -- so that its repair never meets estimates of another launch
    const int slot = lr ? h->lr_next : 0;
    if (lr) {
        if (h->lr_slot[slot].pending) { HIPCHK(h, hipStreamSynchronize(h->stream)); if ((r = lr_check_recover(h, slot))) return r; }
        h->lr_next = (slot + 1) % dvbs2hip_handle::LR_SLOTS;
    }
    if ((r = ensure(h, lr ? B_LR_TMP0 + slot : B_SFF_TMP, sizeof(float) * 4 * (size_t)F, &tmp))) return r;
    const float *x = X_N1;
    float *y = Y_N2, *frq = FRQ, *phs = PHS;
    if (host) {
        if ((r = lr_check_all(h))) return r;          // (the host form reuses B_IN / B_OUT: nothing of an earlier device-form call is left pending behind it)
        if ((r = ensure(h, B_IN, nb, &din)) || (r = ensure(h, B_OUT, nb, &dout)) || (r = ensure(h, B_SFF_OUT, sizeof(float) * 2 * (size_t)F, &dfp))) return r;
        HIPCHK(h, hipMemcpyAsync(din, X_N1, nb, hipMemcpyHostToDevice, h->stream));
        x = (const float *)din; y = (float *)dout; frq = (float *)dfp; phs = frq + F;
    }
    {
        Timer tm(h, DVBS2HIP_K_MISC);
        if (lr) HIPCHK(h, sff_lr_launch(x, y, h->d_lr_R, (float *)tmp, frq, phs, n, F, h->lr_alpha, h->lr_err_dev + slot, h->stream));
        else HIPCHK(h, sff_fp_launch(x, y, (float *)tmp, frq, phs, n, F, h->stream));
    }
    if (lr) { auto &ls = h->lr_slot[slot]; ls.x = x; ls.y = y; ls.n = n; ls.F = F; ls.pending = !host; }
    if (host) {
        HIPCHK(h, hipMemcpyAsync(Y_N2, dout, nb, hipMemcpyDeviceToHost, h->stream));
        if (FRQ) HIPCHK(h, hipMemcpyAsync(FRQ, frq, sizeof(float) * (size_t)F, hipMemcpyDeviceToHost, h->stream));
        if (PHS) HIPCHK(h, hipMemcpyAsync(PHS, phs, sizeof(float) * (size_t)F, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (lr && h->lr_err_host && ((volatile uint32_t *)h->lr_err_host)[slot]) {      // timeout in the fused L&R launch: rotate again, copy again
            if ((r = lr_check_recover(h, slot))) return r;
            HIPCHK(h, hipMemcpyAsync(Y_N2, dout, nb, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
    }
    return 0;
}

int dvbs2hip_sync_lr_synchronize(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t F) { return sff_call(h, true, true, X_N1, FRQ, PHS, Y_N2, F); }
int dvbs2hip_sync_lr_synchronize_dev(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t F) { return sff_call(h, true, false, X_N1, FRQ, PHS, Y_N2, F); }
int dvbs2hip_sync_freq_phase_synchronize(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t F) { return sff_call(h, false, true, X_N1, FRQ, PHS, Y_N2, F); }
int dvbs2hip_sync_freq_phase_synchronize_dev(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t F) { return sff_call(h, false, false, X_N1, FRQ, PHS, Y_N2, F); }

int dvbs2hip_sync_lr_set_alpha(dvbs2hip_t *h, float alpha)
{
    if (!h) return DVBS2HIP_EINVAL;
    h->lr_alpha = alpha;
    return 0;
}

int dvbs2hip_sync_lr_timeouts(dvbs2hip_t *h, int32_t *n)
{
    if (!h || !n) return DVBS2HIP_EINVAL;
    *n = h->lr_timeouts;
    return 0;
}

int dvbs2hip_sync_lr_reset(dvbs2hip_t *h)          // Synchronizer_Luise_Reggiannini_DVBS2_aib::_reset, .cpp:170-176
{
    if (!h) return DVBS2HIP_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, DVBS2HIP_EHIP, "hipSetDevice failed");
    if (h->d_lr_R) HIPCHK(h, hipMemsetAsync(h->d_lr_R, 0, 2 * sizeof(float), h->stream));
    return 0;
}

int dvbs2hip_sync_frame_get_metric(dvbs2hip_t *h, float *max_corr, int32_t *packet_flag)
{
    if (!h || !max_corr || !packet_flag) return DVBS2HIP_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, DVBS2HIP_EHIP, "hipSetDevice failed");
    int r = sfm_ready(h); if (r) return r;
    float m = 0.f;
    HIPCHK(h, hipMemcpyAsync(&m, h->sfm.metric, sizeof m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *max_corr = m; *packet_flag = m > h->sfm.trigger ? 1 : 0;                      // _get_metric / _get_packet_flag, .hpp:59-60
    return 0;
}

// ------------------------------------------------------------------ a6
int dvbs2hip_estimate_dev(dvbs2hip_t *h, const float *X, float *SIG, float *EB, float *ES, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X || !SIG || !EB || !ES) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, estimate_launch(X, SIG, EB, ES, h->n_sym, h->code_rate, h->bps, F, h->stream));
    return 0;
}

int dvbs2hip_estimate(dvbs2hip_t *h, const float *X, float *SIG, float *EB, float *ES, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X || !SIG || !EB || !ES) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t nin = (size_t)F * 2 * h->n_sym * 4;
    void *din, *dout;
    if ((r = ensure(h, B_IN, nin, &din)) || (r = ensure(h, B_OUT, (size_t)F * 12, &dout))) return r;
    float *o = (float *)dout;
    HIPCHK(h, hipMemcpyAsync(din, X, nin, hipMemcpyHostToDevice, h->stream));
    if ((r = dvbs2hip_estimate_dev(h, (const float *)din, o, o + F, o + 2 * F, F))) return r;
    HIPCHK(h, hipMemcpyAsync(SIG, o, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(EB, o + F, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ES, o + 2 * F, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ------------------------------------------------------------------ Multiplier_AGC_cc_naive::imultiply (the two gain stages of the reference's RX graph)
int dvbs2hip_agc_imultiply_dev(dvbs2hip_t *h, const float *X, float *Z, int32_t n_cplx, float output_energy, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X || !Z) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    if (n_cplx < 1) return fail(h, DVBS2HIP_EINVAL, "'n_cplx' has to be greater than 0");
    if (!(output_energy > 0.f)) return fail(h, DVBS2HIP_EINVAL, "'output_energy' has to be greater than 0");
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, agc_launch(X, Z, n_cplx, output_energy, F, h->stream));
    return 0;
}
int dvbs2hip_agc_imultiply(dvbs2hip_t *h, const float *X, float *Z, int32_t n_cplx, float output_energy, int32_t F)
{
    const size_t n = (size_t)2 * (n_cplx > 0 ? n_cplx : 0);
    return host_wrap<true>(h, X, n, Z, n, F, [&](const float *a, float *b, int nf) { return dvbs2hip_agc_imultiply_dev(h, a, b, n_cplx, output_energy, nf); });
}

// ------------------------------------------------------------------ Synchronizer_freq_coarse::synchronize in the transmission phase (the frequency shift; the loop that finds it is sample-serial)
int dvbs2hip_sync_coarse_set_freq(dvbs2hip_t *h, float estimated_freq)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (!(estimated_freq == estimated_freq) || fabsf(estimated_freq) > 0.5f) return fail(h, DVBS2HIP_EINVAL, "'estimated_freq' has to be a normalized frequency in [-0.5, 0.5]");
    const float nu = -estimated_freq;                                  // Synchronizer_freq_coarse_DVBS2_aib.cpp:82: mult.set_nu(-estimated_freq)
    const float new_nu = floorf(nu * 1e6f) / 1e6f;                     // Multiplier_sine_ccc_naive::set_nu, .cpp:44-51
    h->nco_nu = new_nu;
    h->nco_omega = (float)(2 * 3.1415926535897932384626433832795 * new_nu);
    return 0;
}
int dvbs2hip_sync_coarse_reset(dvbs2hip_t *h)                          // Synchronizer_freq_coarse_DVBS2_aib::_reset, .cpp:123-135
{
    if (!h) return DVBS2HIP_EINVAL;
    h->nco_n = 0; h->nco_nu = 0.f; h->nco_omega = 0.f;
    return 0;
}
int dvbs2hip_sync_coarse_synchronize_dev(dvbs2hip_t *h, const float *X, float *FRQ, float *PHS, float *Y, int32_t n_cplx, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!X || !Y) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    if (n_cplx < 1) return fail(h, DVBS2HIP_EINVAL, "'n_cplx' has to be greater than 0");
    if (h->capturing) return fail(h, DVBS2HIP_EUNSUPPORTED, "the stream position is a launch argument: this task cannot be recorded into a graph");
    Timer tm(h, DVBS2HIP_K_MISC);
    const long long total = (long long)n_cplx * F;
    HIPCHK(h, nco_launch(X, Y, h->nco_omega, h->nco_n, total, FRQ, PHS, -h->nco_nu, F, h->stream));
    h->nco_n = (uint32_t)(((unsigned long long)h->nco_n + (unsigned long long)total) % 1000000ull);
    return 0;
}
int dvbs2hip_sync_coarse_synchronize(dvbs2hip_t *h, const float *X, float *FRQ, float *PHS, float *Y, int32_t n_cplx, int32_t F)
{
    const size_t n = (size_t)2 * (n_cplx > 0 ? n_cplx : 0);
    // (one stream in order: the chunks of pinned sockets advance the sample counter as they come)
    int r = host_wrap<true>(h, X, n, Y, n, F, [&](const float *a, float *b, int nf) { return dvbs2hip_sync_coarse_synchronize_dev(h, a, nullptr, nullptr, b, n_cplx, nf); });
    if (r) return r;
    for (int f = 0; f < F; f++) { if (FRQ) FRQ[f] = -h->nco_nu; if (PHS) PHS[f] = 0.f; }
    return 0;
}

// ------------------------------------------------------------------ a7
int dvbs2hip_pl_descramble_dev(dvbs2hip_t *h, const float *a, float *b, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!a || !b) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, pl_descramble_launch(a, b, h->d_pl_seq, h->pl_frame, F, h->stream));
    return 0;
}
int dvbs2hip_pl_descramble(dvbs2hip_t *h, const float *a, float *b, int32_t F)
{
    const size_t n = h ? (size_t)2 * h->pl_frame : 0;
    return host_wrap<true>(h, a, n, b, n, F, [&](const float *x, float *y, int nf) { return dvbs2hip_pl_descramble_dev(h, x, y, nf); });
}
int dvbs2hip_remove_plh_dev(dvbs2hip_t *h, const float *a, float *b, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!a || !b) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, remove_plh_launch(a, b, h->n_sym, h->pl_frame, F, h->stream));
    return 0;
}
int dvbs2hip_remove_plh(dvbs2hip_t *h, const float *a, float *b, int32_t F)
{
    return host_wrap<true>(h, a, h ? (size_t)2 * h->pl_frame : 0, b, h ? (size_t)2 * h->n_sym : 0, F,
                           [&](const float *x, float *y, int nf) { return dvbs2hip_remove_plh_dev(h, x, y, nf); });
}

// ------------------------------------------------------------------ a8
int dvbs2hip_bb_descramble_dev(dvbs2hip_t *h, const int32_t *a, int32_t *b, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!a || !b) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, bb_descramble_launch(a, b, h->bch.d_prbs, h->K_bch, F, h->stream));
    return 0;
}
int dvbs2hip_bb_descramble(dvbs2hip_t *h, const int32_t *a, int32_t *b, int32_t F)
{
    const size_t n = h ? (size_t)h->K_bch : 0;
    return host_wrap<true>(h, a, n, b, n, F, [&](const int32_t *x, int32_t *y, int nf) { return dvbs2hip_bb_descramble_dev(h, x, y, nf); });
}

// ------------------------------------------------------------------ a9
int dvbs2hip_monitor_check_errors_dev(dvbs2hip_t *h, const int32_t *U, const int32_t *V, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!U || !V) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, monitor_launch(U, V, h->d_ctr, h->K_bch, F, h->stream));
    return 0;
}
int dvbs2hip_monitor_check_errors(dvbs2hip_t *h, const int32_t *U, const int32_t *V, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!U || !V) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t n = (size_t)F * h->K_bch * 4;
    void *du, *dv;
    if ((r = ensure(h, B_IN, n, &du)) || (r = ensure(h, B_OUT, n, &dv))) return r;
    HIPCHK(h, hipMemcpyAsync(du, U, n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dv, V, n, hipMemcpyHostToDevice, h->stream));
    if ((r = dvbs2hip_monitor_check_errors_dev(h, (const int32_t *)du, (const int32_t *)dv, F))) return r;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
// check_errors2: the counters after every frame of the call, as sockets (device pointers; any of the five may be NULL)
int dvbs2hip_monitor_check_errors2_dev(dvbs2hip_t *h, const int32_t *U, const int32_t *V, int64_t *FRA, int32_t *BE, int32_t *FE, float *BER, float *FER, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!U || !V) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    void *tmp;
    if ((r = ensure(h, B_MON_BE, (size_t)F * 4, &tmp))) return r;
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, monitor2_launch(U, V, h->d_ctr, (int32_t *)tmp, (long long *)FRA, BE, FE, BER, FER, h->K_bch, F, h->stream));
    return 0;
}
int dvbs2hip_monitor_check_errors2(dvbs2hip_t *h, const int32_t *U, const int32_t *V, int64_t *FRA, int32_t *BE, int32_t *FE, float *BER, float *FER, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!U || !V) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t n = (size_t)F * h->K_bch * 4;
    void *du, *dv, *dout;
    if ((r = ensure(h, B_IN, n, &du)) || (r = ensure(h, B_OUT, n, &dv)) || (r = ensure(h, B_MON_OUT, (size_t)F * 24, &dout))) return r;
    int64_t *dfra = (int64_t *)dout; int32_t *dbe = (int32_t *)(dfra + F), *dfe = dbe + F; float *dber = (float *)(dfe + F), *dfer = dber + F;
    HIPCHK(h, hipMemcpyAsync(du, U, n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dv, V, n, hipMemcpyHostToDevice, h->stream));
    if ((r = dvbs2hip_monitor_check_errors2_dev(h, (const int32_t *)du, (const int32_t *)dv, dfra, dbe, dfe, dber, dfer, F))) return r;
    if (FRA) HIPCHK(h, hipMemcpyAsync(FRA, dfra, (size_t)F * 8, hipMemcpyDeviceToHost, h->stream));
    if (BE) HIPCHK(h, hipMemcpyAsync(BE, dbe, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
    if (FE) HIPCHK(h, hipMemcpyAsync(FE, dfe, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
    if (BER) HIPCHK(h, hipMemcpyAsync(BER, dber, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
    if (FER) HIPCHK(h, hipMemcpyAsync(FER, dfer, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- Monitor_reduction across GPUs (TX_RX_BB/main.cpp:123-125,155-161): one process per GPU, ONE RCCL all-reduce of 3 x uint64.
// librccl is opened at run time (the library itself does not link it: single-GPU users never load it).
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, dvbs2hip_nccl_id, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;      // optional
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;
const char *rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);      // two handles of one process may initialise their reductions from different threads
    if (g_rccl.lib) return nullptr;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *l = nullptr;
    for (const char *n : names) if ((l = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // the copy the process already has (torch ships one)
    if (!l) for (const char *n : names) if ((l = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!l) return "librccl.so not found (dlopen)";
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(l, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void **, int, dvbs2hip_nccl_id, int))dlsym(l, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(l, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(l, "ncclCommDestroy");
    g_rccl.CommAbort = (int (*)(void *))dlsym(l, "ncclCommAbort");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(l, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) return "librccl.so lacks the nccl* entry points";
    g_rccl.lib = l;
    return nullptr;
}
std::string rccl_err(int e) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : ("rccl error " + std::to_string(e)); }
}  // namespace

// The out-of-band step of the RCCL bootstrap (no GPU involved; exported so that CPU processes can exercise it): rank 0 hands `bytes`
// bytes to every other rank through files named after `path`.  A file found under that name is never trusted for being there (a run
// that crashed leaves files behind, and a fixed name under /tmp can be squatted): every reader publishes a fresh random nonce in
// `path.hello.<rank>` and accepts only a `path.ack.<rank>` that carries the same nonce in front of the payload; rank 0 answers every
// hello it sees, answers again when a hello's nonce changes (a stale hello being replaced by the live reader's), and is done with a
// rank once that rank has consumed (removed) its ack.  Nothing is left behind by a successful exchange.  Files are created beside and
// renamed (nobody reads a partial file), with O_EXCL | O_NOFOLLOW and mode 0600; a private directory is still the better place.
namespace {
constexpr size_t RDV_NONCE = 16;
bool rdv_write(const std::string &name, const void *a, size_t na, const void *b, size_t nb)
{
    const std::string tmp = name + ".tmp." + std::to_string((long long)getpid());
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    if (fd < 0) return false;
    bool ok = (size_t)write(fd, a, na) == na && (nb == 0 || (size_t)write(fd, b, nb) == nb);
    ok = close(fd) == 0 && ok;
    if (ok) ok = rename(tmp.c_str(), name.c_str()) == 0;
    if (!ok) (void)unlink(tmp.c_str());
    return ok;
}
bool rdv_read(const std::string &name, void *buf, size_t n)      // true only for a regular file that holds exactly n bytes
{
    const int fd = open(name.c_str(), O_RDONLY | O_NOFOLLOW);
    if (fd < 0) return false;
    std::vector<unsigned char> tmp(n + 1);
    size_t got = 0;
    for (;;) { const ssize_t r = read(fd, tmp.data() + got, n + 1 - got); if (r <= 0) break; got += (size_t)r; if (got > n) break; }
    close(fd);
    if (got != n) return false;
    memcpy(buf, tmp.data(), n);
    return true;
}
void rdv_nonce(unsigned char *n)
{
    bool ok = false;
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) { ok = read(fd, n, RDV_NONCE) == (ssize_t)RDV_NONCE; close(fd); }
    if (!ok) {
        struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
        unsigned long long v[2] = {(unsigned long long)ts.tv_nsec ^ ((unsigned long long)getpid() << 32), (unsigned long long)ts.tv_sec ^ (unsigned long long)(uintptr_t)n};
        memcpy(n, v, RDV_NONCE);
    }
}
}  // namespace

int dvbs2hip_rendezvous(int32_t rank, int32_t world, const char *path, void *blob, size_t bytes, int32_t timeout_ms)
{
    if (world < 1 || rank < 0 || rank >= world || !path || !*path || !blob || !bytes) return DVBS2HIP_EINVAL;
    if (world == 1) return 0;
    const std::string base(path);
    const int step_ms = 20;
    if (rank > 0) {
        unsigned char nonce[RDV_NONCE];
        rdv_nonce(nonce);
        const std::string hello = base + ".hello." + std::to_string(rank), ack = base + ".ack." + std::to_string(rank);
        if (!rdv_write(hello, nonce, RDV_NONCE, nullptr, 0)) return fail(nullptr, DVBS2HIP_EINVAL, "cannot write the rendezvous file " + hello);
        std::vector<unsigned char> buf(RDV_NONCE + bytes);
        for (int waited = 0;; waited += step_ms) {
            if (rdv_read(ack, buf.data(), buf.size()) && !memcmp(buf.data(), nonce, RDV_NONCE)) {
                memcpy(blob, buf.data() + RDV_NONCE, bytes);
                (void)unlink(ack.c_str()); (void)unlink(hello.c_str());
                return 0;
            }
            if (timeout_ms >= 0 && waited >= timeout_ms) { (void)unlink(hello.c_str()); return fail(nullptr, DVBS2HIP_EHIP, "timed out waiting for rank 0 at the rendezvous " + base); }
            usleep(step_ms * 1000);
        }
    }
    std::vector<char> acked(world, 0), done(world, 0);
    std::vector<unsigned char> nonces((size_t)world * RDV_NONCE, 0);
    int left = world - 1;
    for (int waited = 0; left > 0; waited += step_ms) {
        for (int r = 1; r < world; r++) {
            if (done[r]) continue;
            const std::string hello = base + ".hello." + std::to_string(r), ack = base + ".ack." + std::to_string(r);
            unsigned char n[RDV_NONCE];
            if (rdv_read(hello, n, RDV_NONCE) && (!acked[r] || memcmp(n, &nonces[(size_t)r * RDV_NONCE], RDV_NONCE))) {
                if (!rdv_write(ack, n, RDV_NONCE, blob, bytes)) return fail(nullptr, DVBS2HIP_EINVAL, "cannot write the rendezvous file " + ack);
                memcpy(&nonces[(size_t)r * RDV_NONCE], n, RDV_NONCE); acked[r] = 1;
            } else if (acked[r] && access(ack.c_str(), F_OK) != 0) { done[r] = 1; left--; }      // consumed by its reader
        }
        if (left > 0) {
            if (timeout_ms >= 0 && waited >= timeout_ms) {
                for (int r = 1; r < world; r++) if (acked[r] && !done[r]) (void)unlink((base + ".ack." + std::to_string(r)).c_str());
                return fail(nullptr, DVBS2HIP_EHIP, "timed out waiting for " + std::to_string(left) + " rank(s) at the rendezvous " + base);
            }
            usleep(step_ms * 1000);
        }
    }
    return 0;
}

int dvbs2hip_monitor_reduce_init(dvbs2hip_t *h, int32_t rank, int32_t world, const char *rendezvous, int32_t timeout_ms)
{
    int r0 = enter(h); if (r0) return r0;
    if (world < 1 || rank < 0 || rank >= world) return fail(h, DVBS2HIP_EINVAL, "'rank' has to be in [0, world_size)");
    if (h->nccl_comm) return fail(h, DVBS2HIP_EINVAL, "the monitor reduction is already initialised on this handle");
    if (world > 1 && (!rendezvous || !*rendezvous)) return fail(h, DVBS2HIP_EINVAL, "a rendezvous file path is needed for world_size > 1");
    if (const char *e = rccl_load()) return fail(h, DVBS2HIP_EUNSUPPORTED, e);
    dvbs2hip_nccl_id id;
    memset(&id, 0, sizeof id);
    if (rank == 0) {
        int e = g_rccl.GetUniqueId(&id);
        if (e) return fail(h, DVBS2HIP_EHIP, "ncclGetUniqueId: " + rccl_err(e));
    }
    if (world > 1) {
        const int rr = dvbs2hip_rendezvous(rank, world, rendezvous, &id, sizeof id, timeout_ms);
        if (rr) return fail(h, rr, dvbs2hip_last_error(nullptr));
    }
    void *comm = nullptr;
    int e = g_rccl.CommInitRank(&comm, world, id, rank);
    if (e) return fail(h, DVBS2HIP_EHIP, "ncclCommInitRank: " + rccl_err(e));
    if (!h->d_red && hipMalloc((void **)&h->d_red, 3 * sizeof(unsigned long long)) != hipSuccess) { (void)g_rccl.CommDestroy(comm); return fail(h, DVBS2HIP_ENOMEM, "hipMalloc failed"); }
    if (!h->h_red && hipHostMalloc((void **)&h->h_red, 3 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { h->h_red = nullptr; (void)g_rccl.CommDestroy(comm); return fail(h, DVBS2HIP_ENOMEM, "hipHostMalloc failed"); }
    h->nccl_comm = comm; h->red_rank = rank; h->red_world = world; h->red_timeout_ms = timeout_ms > 0 ? timeout_ms : 0;
    return 0;
}

int dvbs2hip_monitor_reduce(dvbs2hip_t *h, uint64_t out[3])
{
    if (!h || !out) return DVBS2HIP_EINVAL;
    if (!h->nccl_comm) return dvbs2hip_monitor_get(h, out);         // a single process: the local counters are the sum
    int r0 = enter(h); if (r0) return r0;
    const int e = g_rccl.AllReduce(h->d_ctr, h->d_red, 3, 5 /* ncclUint64 */, 0 /* ncclSum */, h->nccl_comm, h->stream);
    if (e) return fail(h, DVBS2HIP_EHIP, "ncclAllReduce: " + rccl_err(e));
    unsigned long long *tmp = h->h_red;
    HIPCHK(h, hipMemcpyAsync(tmp, h->d_red, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    // A peer that died leaves this rank's all-reduce waiting on the device for ever: wait with the timeout given at _reduce_init instead of a blocking
    // synchronize -- a launcher-less `dvbs2_tx_rx_bb --world N` then ends with a non-zero exit code of its own instead of hanging until somebody kills it.
    if (h->red_timeout_ms > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIPCHK(h, q);
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > h->red_timeout_ms) {
                if (g_rccl.CommAbort) (void)g_rccl.CommAbort(h->nccl_comm);
                h->nccl_comm = nullptr; h->comm_dead = true;
                return fail(h, DVBS2HIP_ETIMEOUT, "monitor reduction: a peer rank did not arrive within " + std::to_string(h->red_timeout_ms) + " ms (rank " + std::to_string(h->red_rank) + " of " + std::to_string(h->red_world) + ")");
            }
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= 5) std::this_thread::sleep_for(std::chrono::microseconds(50));      // (spin for the first 5 ms: a reduction that arrives is microseconds away)
        }
    } else
        HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 3; i++) out[i] = tmp[i];
    return 0;
}

int dvbs2hip_monitor_reduce_finalize(dvbs2hip_t *h)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (h->nccl_comm) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        (void)g_rccl.CommDestroy(h->nccl_comm);
        h->nccl_comm = nullptr; h->red_world = 1; h->red_rank = 0;
    }
    return 0;
}

int dvbs2hip_monitor_get(dvbs2hip_t *h, uint64_t out[3])
{
    if (!h || !out) return DVBS2HIP_EINVAL;
    int r0 = enter(h); if (r0) return r0;
    unsigned long long tmp[3];
    HIPCHK(h, hipMemcpyAsync(tmp, h->d_ctr, sizeof tmp, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 3; i++) out[i] = tmp[i];
    return 0;
}
int dvbs2hip_monitor_reset(dvbs2hip_t *h)
{
    int r0 = enter(h); if (r0) return r0;
    HIPCHK(h, hipMemsetAsync(h->d_ctr, 0, 3 * sizeof(unsigned long long), h->stream));
    return 0;
}

// ------------------------------------------------------------------ fused RX baseband chain
static int rx_bb_any_dev(dvbs2hip_t *h, const float *pl, const float *const *src, const float *sigma, int32_t *info, int8_t *cwd_l, int8_t *cwd_b, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if ((!pl && !src) || !info) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    void *dllr, *dpk, *dest;
    const size_t nwords = (size_t)(h->K_ldpc + 31) / 32;
    if ((r = ensure(h, B_LLR, (size_t)F * h->N_ldpc * 4, &dllr)) || (r = ensure(h, B_PACKED, (size_t)F * nwords * 4, &dpk)) ||
        (r = ensure(h, B_EST, (size_t)F * 12, &dest)))
        return r;
    {
        Timer tm(h, DVBS2HIP_K_FRONT);
        FrontKParams fp = front_params(h, pl, sigma, (float *)dllr, (float *)dest, F);
        fp.src = src;
        HIPCHK(h, front_rx_launch(fp, h->stream));
    }
    // the LDPC kernel writes the descrambled info bits of every frame straight into the output socket (what the BCH stage outputs for a
    // frame it does not correct: nearly all of them behind a converged LDPC decoder); the BCH stage then only checks the syndromes of the
    // packed hard decisions and flips the bits it corrects -- its 4 K_bch output bytes per frame were 90 % of its time
    const bool fused_out = ldpc_writes_info(h) && !ldpc_lat_ok(h, F);      // (the latency kernel of small batches writes the packed hard decisions only: the BCH stage does the rest)
    // (round 5) ... and forms the frame's BCH remainder r(x) mod g(x) from the hard decisions it outputs: a frame whose remainder is zero is finished (information bits and
    // both CWD flags written by the LDPC kernel); the BCH stage rebuilds and decodes the flagged frames only
    void *dflag = nullptr;
    if (fused_out && ldpc_verifies_bch(h) && (r = ensure(h, B_BCHFLAG, (size_t)F, &dflag))) return r;
    if ((r = ldpc_dev(h, (const float *)dllr, cwd_l, nullptr, (uint32_t *)dpk, nullptr, nullptr, F, fused_out ? info : nullptr, (uint8_t *)dflag, cwd_b))) return r;
    return bch_dev(h, nullptr, (const uint32_t *)dpk, cwd_b, info, true, F, fused_out, (const uint8_t *)dflag);
}

int dvbs2hip_rx_bb_dev(dvbs2hip_t *h, const float *pl, const float *sigma, int32_t *info, int8_t *cwd_l, int8_t *cwd_b, int32_t F)
{
    if (h && !pl) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    return rx_bb_any_dev(h, pl, nullptr, sigma, info, cwd_l, cwd_b, F);
}

// (round 5) the fused chain behind dvbs2hip_sync_frame_locate_dev: frame f is read where SRC[f] points (device table of device pointers)
int dvbs2hip_rx_bb_located_dev(dvbs2hip_t *h, const float *const *SRC, const float *sigma, int32_t *info, int8_t *cwd_l, int8_t *cwd_b, int32_t F)
{
    if (h && !SRC) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    return rx_bb_any_dev(h, nullptr, SRC, sigma, info, cwd_l, cwd_b, F);
}

int dvbs2hip_rx_bb(dvbs2hip_t *h, const float *pl, const float *sigma, int32_t *info, int8_t *cwd_l, int8_t *cwd_b, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!pl || !info) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t nin = (size_t)F * 2 * h->pl_frame * 4, nout = (size_t)F * h->K_bch * 4;
    void *din, *dout, *dc0, *dc1, *dsig = nullptr;
    if ((r = ensure(h, B_IN, nin, &din)) || (r = ensure(h, B_INFO, nout, &dout)) || (r = ensure(h, B_CWD0, F, &dc0)) ||
        (r = ensure(h, B_CWD1, F, &dc1)))
        return r;
    if (sigma) {
        if ((r = ensure(h, B_SIG, (size_t)F * 4, &dsig))) return r;
        HIPCHK(h, hipMemcpyAsync(dsig, sigma, (size_t)F * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (host_is_pinned(h, pl, nin) && host_is_pinned(h, info, nout) && (!cwd_l || host_is_pinned(h, cwd_l, (size_t)F)) &&
        (!cwd_b || host_is_pinned(h, cwd_b, (size_t)F))) {
        const size_t P2 = (size_t)2 * h->pl_frame, K = (size_t)h->K_bch;
        return host_pipeline(h, F, {{pl, din, P2 * 4}}, {{dout, info, K * 4}, {dc0, cwd_l, 1}, {dc1, cwd_b, 1}}, [&](int f0, int nf) {
            return dvbs2hip_rx_bb_dev(h, (const float *)din + (size_t)f0 * P2, dsig ? (const float *)dsig + f0 : nullptr, (int32_t *)dout + (size_t)f0 * K,
                                      (int8_t *)dc0 + f0, (int8_t *)dc1 + f0, nf);
        });
    }
    HIPCHK(h, hipMemcpyAsync(din, pl, nin, hipMemcpyHostToDevice, h->stream));
    if ((r = dvbs2hip_rx_bb_dev(h, (const float *)din, (const float *)dsig, (int32_t *)dout, (int8_t *)dc0, (int8_t *)dc1, F))) return r;
    HIPCHK(h, hipMemcpyAsync(info, dout, nout, hipMemcpyDeviceToHost, h->stream));
    if (cwd_l) HIPCHK(h, hipMemcpyAsync(cwd_l, dc0, F, hipMemcpyDeviceToHost, h->stream));
    if (cwd_b) HIPCHK(h, hipMemcpyAsync(cwd_b, dc1, F, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ------------------------------------------------------------------ N1: TX mirror + AWGN
int dvbs2hip_tx_bb_dev(dvbs2hip_t *h, const int32_t *info_in, uint64_t seed, const float *sigma, int32_t *info_out, float *pl, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!pl) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    void *dbch, *dldpc;
    if ((r = ensure(h, B_TXBCH, (size_t)F * ((h->K_ldpc + 31) / 32) * 4, &dbch)) ||
        (r = ensure(h, B_TXLDPC, (size_t)F * ((h->N_ldpc + 31) / 32) * 4, &dldpc)))
        return r;
    TxKParams p;
    memset(&p, 0, sizeof p);
    p.info_in = info_in; p.info_out = info_out; p.sigma = sigma; p.pl_out = pl;
    p.bch_cw = (uint32_t *)dbch; p.ldpc_cw = (uint32_t *)dldpc; p.prbs = h->bch.d_prbs;
    p.enc_tab = h->d_enc_tab; p.enc_deg = h->d_enc_deg; p.cstl = h->d_cstl; p.plh = h->d_plh; p.pl_seq = h->d_pl_seq;
    p.bch_tab = h->d_bch_tab; p.bch_shift = h->d_bch_shift;
    p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
    p.K_bch = h->K_bch; p.K_ldpc = h->K_ldpc; p.N_ldpc = h->N_ldpc; p.bps = h->bps; p.itl_cols = h->itl_cols; p.itl_order = h->itl_order;
    p.n_sym = h->n_sym; p.pl_frame = h->pl_frame; p.enc_stride = h->enc_stride; p.n_frames = F;
    Timer tm(h, DVBS2HIP_K_MISC);
    HIPCHK(h, tx_launch(p, h->stream));
    return 0;
}

int dvbs2hip_tx_bb(dvbs2hip_t *h, const int32_t *info_in, uint64_t seed, const float *sigma, int32_t *info_out, float *pl, int32_t F)
{
    int r = check_frames(h, F); if (r) return r;
    if (!pl) return fail(h, DVBS2HIP_EINVAL, "null socket pointer");
    const size_t nb = (size_t)F * h->K_bch * 4, npl = (size_t)F * 2 * h->pl_frame * 4;
    void *din = nullptr, *dinfo, *dpl, *dsig = nullptr;
    if ((r = ensure(h, B_INFO, nb, &dinfo)) || (r = ensure(h, B_OUT, npl, &dpl))) return r;
    if (info_in) { if ((r = ensure(h, B_IN, nb, &din))) return r; HIPCHK(h, hipMemcpyAsync(din, info_in, nb, hipMemcpyHostToDevice, h->stream)); }
    if (sigma) { if ((r = ensure(h, B_SIG, (size_t)F * 4, &dsig))) return r; HIPCHK(h, hipMemcpyAsync(dsig, sigma, (size_t)F * 4, hipMemcpyHostToDevice, h->stream)); }
    if ((r = dvbs2hip_tx_bb_dev(h, (const int32_t *)din, seed, (const float *)dsig, (int32_t *)dinfo, (float *)dpl, F))) return r;
    if (info_out) HIPCHK(h, hipMemcpyAsync(info_out, dinfo, nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(pl, dpl, npl, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ------------------------------------------------------------------ measurement + memory helpers
int dvbs2hip_timing_enable(dvbs2hip_t *h, int32_t on)
{
    if (!h) return DVBS2HIP_EINVAL;
    if (h->capturing) return fail(h, DVBS2HIP_EINVAL, "a capture is open on this handle");
    h->timing = on != 0;
    return 0;
}
int dvbs2hip_timing_reset(dvbs2hip_t *h)
{
    if (!h) return DVBS2HIP_EINVAL;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < DVBS2HIP_K_COUNT; k++) {
        for (auto &p : h->ev[k]) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        h->ev[k].clear();
    }
    return 0;
}
int dvbs2hip_timing_get(dvbs2hip_t *h, int32_t k, double *total_ms, int64_t *n)
{
    if (!h || k < 0 || k >= DVBS2HIP_K_COUNT || !total_ms || !n) return DVBS2HIP_EINVAL;
    int r0 = enter(h); if (r0) return r0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double tot = 0.0;
    for (auto &p : h->ev[k]) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, p.first, p.second));
        tot += ms;
    }
    *total_ms = tot; *n = (int64_t)h->ev[k].size();
    return 0;
}
// a plain streaming copy, 16 bytes per lane and access, U accesses in flight per lane; NT: non-temporal loads and stores.  Flat grid: workgroup b copies the
// U consecutive 4 KB pieces starting at piece b U (what the front end and the synchronizers' rotation do with their streams).
typedef float copy_f4 __attribute__((ext_vector_type(4)));
extern "C++" {
template <int U, bool NT>
__global__ void __launch_bounds__(256) device_copy_kernel(const copy_f4 *__restrict__ src, copy_f4 *__restrict__ dst, size_t n4)
{
    const size_t i0 = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    copy_f4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) { const size_t i = i0 + (size_t)k * 256; if (i < n4) v[k] = NT ? __builtin_nontemporal_load(&src[i]) : src[i]; }
#pragma unroll
    for (int k = 0; k < U; k++) { const size_t i = i0 + (size_t)k * 256; if (i < n4) { if (NT) __builtin_nontemporal_store(v[k], &dst[i]); else dst[i] = v[k]; } }
}
}  // extern "C++"

int dvbs2hip_device_copy_bandwidth(dvbs2hip_t *h, size_t bytes, int32_t reps, double *GBps)
{
    if (!h || !GBps || reps < 1 || bytes < 16) return DVBS2HIP_EINVAL;
    int r0 = enter(h); if (r0) return r0;
    const size_t n4 = bytes / 16;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, n4 * 16) != hipSuccess) return fail(h, DVBS2HIP_ENOMEM, "hipMalloc of " + std::to_string(n4 * 16) + " bytes failed");
    if (hipMalloc(&b, n4 * 16) != hipSuccess) { (void)hipFree(a); return fail(h, DVBS2HIP_ENOMEM, "hipMalloc of " + std::to_string(n4 * 16) + " bytes failed"); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    double best = 0.0;
    if (hipMemsetAsync(a, 0x3c, n4 * 16, h->stream) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = DVBS2HIP_EHIP;
    // the best of four shapes (1 / 4 accesses in flight per lane, plain / non-temporal): which one wins depends on the size against the Infinity Cache
    for (int shape = 0; shape < 4 && !rc; shape++) {
        const int U = shape & 1 ? 4 : 1;
        const unsigned grid = (unsigned)((n4 + (size_t)256 * U - 1) / ((size_t)256 * U));
        auto launch = [&]() {
            if (shape == 0) hipLaunchKernelGGL((device_copy_kernel<1, false>), dim3(grid), dim3(256), 0, h->stream, (const copy_f4 *)a, (copy_f4 *)b, n4);
            else if (shape == 1) hipLaunchKernelGGL((device_copy_kernel<4, false>), dim3(grid), dim3(256), 0, h->stream, (const copy_f4 *)a, (copy_f4 *)b, n4);
            else if (shape == 2) hipLaunchKernelGGL((device_copy_kernel<1, true>), dim3(grid), dim3(256), 0, h->stream, (const copy_f4 *)a, (copy_f4 *)b, n4);
            else hipLaunchKernelGGL((device_copy_kernel<4, true>), dim3(grid), dim3(256), 0, h->stream, (const copy_f4 *)a, (copy_f4 *)b, n4);
        };
        float ms = 0.f;
        launch();      // warm-up
        (void)hipEventRecord(e0, h->stream);
        for (int i = 0; i < reps; i++) launch();
        (void)hipEventRecord(e1, h->stream);
        if (hipStreamSynchronize(h->stream) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) rc = DVBS2HIP_EHIP;
        else { const double g = 2.0 * (double)(n4 * 16) * reps / ((double)ms * 1e-3) / 1e9; if (g > best) best = g; }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    if (rc) return fail(h, rc, "device copy measurement failed");
    *GBps = best;
    return 0;
}

int dvbs2hip_malloc(dvbs2hip_t *h, void **d, size_t bytes)
{
    if (!h || !d) return DVBS2HIP_EINVAL;
    int r0 = enter(h); if (r0) return r0;
    if (hipMalloc(d, bytes) != hipSuccess) return fail(h, DVBS2HIP_ENOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
    return 0;
}
int dvbs2hip_free(dvbs2hip_t *h, void *d) { int r0 = enter(h); if (r0) return r0; HIPCHK(h, hipFree(d)); return 0; }
int dvbs2hip_memcpy_h2d(dvbs2hip_t *h, void *dst, const void *src, size_t bytes)
{
    int r0 = enter(h); if (r0) return r0;
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int dvbs2hip_memcpy_d2h(dvbs2hip_t *h, void *dst, const void *src, size_t bytes)
{
    int r0 = enter(h); if (r0) return r0;
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

}  // extern "C"
