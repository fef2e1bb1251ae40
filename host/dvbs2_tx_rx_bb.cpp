// dvbs2_tx_rx_bb (HIP) -- the reference's Monte-Carlo BER/FER simulator
// (/root/reference src/mains/TX_RX_BB/main.cpp:28-192) with TX, channel, RX and monitor on one MI355X,
// written against the C ABI only (include/dvbs2hip.h): device buffers from dvbs2hip_malloc, the
// `_dev` entry points chained on the handle's stream, one 24-byte counter read per batch.
// Same flags where they apply to this path (src/common/Factory/DVBS2/DVBS2.cpp:117-149), same table
// as refs/TX_RX_BB/*.txt.
//
//   dvbs2_tx_rx_bb --mod-cod QPSK-S_8/9 -m 3.6 -M 3.81 -s 0.1 --dec-implem SPA --dec-ite 50 -F 2048
//
// Multi-GPU: ONE PROCESS PER GPU.  Start N copies with --world N --rank r --rendezvous /path/to/file (or with the
// environment a torchrun-style launcher sets: WORLD_SIZE, RANK, LOCAL_RANK; the file then defaults to
// /tmp/dvbs2hip_rdv_$MASTER_PORT).  Rank r uses GPU LOCAL_RANK (default r) and its own noise / payload stream
// (seed = base + rank, like the reference's per-clone seeds, main.cpp:118-120); the monitors are reduced by
// dvbs2hip_monitor_reduce = one RCCL all-reduce of {FRA, BE, FE} per batch, which is what tools::Monitor_reduction does
// across the reference's threads (main.cpp:123-125,155-161); every rank stops on the REDUCED frame-error count; rank 0
// prints the table.  A rank that fails leaves with a non-zero exit code; the collectives of the others do not time out, so the
// launcher has to end the job when one process dies (torchrun does).
//
// Clones: the reference runs its chain in hardware_concurrency() clones, each with its own -F frames in flight (main.cpp:19,96 -- spu::runtime::Sequence with
// n_threads; seeds per clone main.cpp:118-120; the monitors summed by Monitor_reduction, main.cpp:123-125).  --clones C (default 3) is that on one GPU: C handles = C
// streams, batches dealt to them in turn, the host waiting on a clone's counters only when that clone's turn comes again.  With the syndrome early stop a batch of the
// LDPC kernel ends with a few frames running to the iteration limit on a few CUs; the next clone's kernels fill the others (one clone: --clones 1, the loop of rounds 2-4).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
#include "../include/dvbs2hip.h"

// (a reduction that timed out leaves GPU work that will never finish behind it: leave at once, without the runtime's teardown -- a fresh exit, never a re-exec)
#define CHK(call) do { int rc__ = (call); if (rc__) { std::fprintf(stderr, "%s: %s\n", #call, dvbs2hip_last_error(h)); std::fflush(stdout); std::fflush(stderr); if (rc__ == DVBS2HIP_ETIMEOUT) std::_Exit(4); return 3; } } while (0)

int main(int argc, char **argv)
{
    std::string modcod = "QPSK-S_8/9", implem = "SPA", est = "DVBS2", sched = "QC";      // --dec-sched NATURAL: the reference's sweep order (k_ldpc_nat.hip; wants -F 32768)
    double ebn0_min = 3.2, ebn0_max = 6.0, step = 0.1;     // DVBS2.cpp:121-123
    int F = 512, n_ite = 50, max_fe = 100, n_clones = 3;   // DVBS2.cpp:135-142 (implem SPA, 50 ite, 100 frame errors)
    long long max_frames = 10000000;
    int reduce_timeout_ms = 120000;      // rendezvous and every later reduction: a rank whose peers do not arrive leaves with exit code 4
    bool sim_stats = false, src_no_loop = false;
    std::string src_type = "RAND", src_path;      // DVBS2.cpp:66,359-376: RAND = the TX mirror's own generator on the device; USER / USER_BIN / AZCW: payloads made here, handed to its `info_in` socket
    float alpha = 1.0f;
    auto env_int = [](const char *n, int d) { const char *v = std::getenv(n); return v && *v ? std::atoi(v) : d; };
    int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local_rank = env_int("LOCAL_RANK", -1);
    std::string rendezvous = std::getenv("MASTER_PORT") ? std::string("/tmp/dvbs2hip_rdv_") + std::getenv("MASTER_PORT") : "";
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--mod-cod") modcod = next();
        else if (a == "-m" || a == "--sim-noise-min") ebn0_min = std::atof(next());
        else if (a == "-M" || a == "--sim-noise-max") ebn0_max = std::atof(next());
        else if (a == "-s" || a == "--sim-noise-step") step = std::atof(next());
        else if (a == "-e" || a == "--max-fe") max_fe = std::atoi(next());
        else if (a == "-F" || a == "--sim-inter-fra") F = std::atoi(next());
        else if (a == "--dec-ite") n_ite = std::atoi(next());
        else if (a == "--dec-implem") implem = next();
        else if (a == "--dec-alpha") alpha = (float)std::atof(next());
        else if (a == "--dec-sched") sched = next();
        else if (a == "--est-type") est = next();
        else if (a == "--max-frames") max_frames = std::atoll(next());
        else if (a == "--clones") n_clones = std::atoi(next());
        else if (a == "--world") world = std::atoi(next());
        else if (a == "--rank") rank = std::atoi(next());
        else if (a == "--local-rank") local_rank = std::atoi(next());
        else if (a == "--rendezvous") rendezvous = next();
        else if (a == "--reduce-timeout-ms") reduce_timeout_ms = std::atoi(next());
        else if (a == "--src-type") src_type = next();
        else if (a == "--src-path") src_path = next();
        else if (a == "--src-no-loop") src_no_loop = true;
        else if (a == "--ter-freq") (void)next();                               // accepted and ignored: a row is printed when its noise point is done
        else if (a == "--sim-stats") sim_stats = true;                         // TX_RX_BB/main.cpp:110,170-178: per-task statistics at the end
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    dvbs2hip_t *h = nullptr;
    dvbs2hip_cfg cfg;
    if (dvbs2hip_cfg_from_modcod(modcod.c_str(), &cfg)) { std::fprintf(stderr, "%s\n", dvbs2hip_last_error(nullptr)); return 3; }
    cfg.max_frames = F; cfg.ldpc_n_ite = n_ite; cfg.ldpc_alpha = alpha; cfg.ldpc_early_stop = 1;
    int32_t n_gpus = 0;
    (void)dvbs2hip_device_count(&n_gpus);
    cfg.device = local_rank >= 0 ? local_rank : (n_gpus > 0 ? rank % n_gpus : rank);      // a global rank only: ranks are dealt to the node's GPUs in order
    cfg.ldpc_implem = implem == "SPA" ? DVBS2HIP_IMPLEM_SPA : implem == "SPA_TANH" ? DVBS2HIP_IMPLEM_SPA_TANH : implem == "SPA_EXACT" ? DVBS2HIP_IMPLEM_SPA_EXACT : implem == "MS" ? DVBS2HIP_IMPLEM_MS : DVBS2HIP_IMPLEM_NMS;
    if (implem != "SPA" && implem != "SPA_TANH" && implem != "SPA_EXACT" && implem != "MS" && implem != "NMS") { std::fprintf(stderr, "--dec-implem has to be SPA, SPA_TANH, SPA_EXACT, MS or NMS\n"); return 2; }
    if (n_clones < 1 || n_clones > 8) { std::fprintf(stderr, "--clones has to be 1 .. 8\n"); return 2; }
    struct Clone { dvbs2hip_t *h = nullptr; void *d_pl = nullptr, *d_sent = nullptr, *d_got = nullptr, *d_sig = nullptr, *d_info = nullptr; uint64_t c[3] = {0, 0, 0}; bool busy = false; };
    if (src_type != "RAND" && src_type != "USER" && src_type != "USER_BIN" && src_type != "AZCW") { std::fprintf(stderr, "--src-type has to be RAND, USER, USER_BIN or AZCW\n"); return 2; }
    if ((src_type == "USER" || src_type == "USER_BIN") && src_path.empty()) { std::fprintf(stderr, "--src-type %s needs --src-path\n", src_type.c_str()); return 2; }
    std::vector<Clone> cl((size_t)n_clones);
    dvbs2hip_sizes sz;
    const bool reduce = world > 1 || std::getenv("DVBS2HIP_FORCE_RCCL");
    for (int k = 0; k < n_clones; k++) {
        if (dvbs2hip_create(&cfg, &cl[k].h)) { std::fprintf(stderr, "%s\n", dvbs2hip_last_error(nullptr)); return 3; }
        h = cl[k].h;
        if (sched == "NATURAL") CHK(dvbs2hip_set_ldpc_schedule(h, DVBS2HIP_SCHED_NATURAL));
        else if (sched != "QC") { std::fprintf(stderr, "--dec-sched has to be QC or NATURAL\n"); return 2; }
        CHK(dvbs2hip_get_sizes(h, &sz));
        if (sim_stats) CHK(dvbs2hip_timing_enable(h, 1));
        // one communicator per clone (its all-reduce runs on the clone's stream); every rank calls them in the same order
        if (reduce) CHK(dvbs2hip_monitor_reduce_init(h, rank, world, (rendezvous + (k ? ".c" + std::to_string(k) : "")).c_str(), reduce_timeout_ms));
        CHK(dvbs2hip_malloc(h, &cl[k].d_pl, (size_t)F * 2 * sz.pl_frame_sym * sizeof(float)));
        CHK(dvbs2hip_malloc(h, &cl[k].d_sent, (size_t)F * sz.K_bch * sizeof(int32_t)));
        CHK(dvbs2hip_malloc(h, &cl[k].d_got, (size_t)F * sz.K_bch * sizeof(int32_t)));
        CHK(dvbs2hip_malloc(h, &cl[k].d_sig, (size_t)F * sizeof(float)));
        if (src_type != "RAND") CHK(dvbs2hip_malloc(h, &cl[k].d_info, (size_t)F * sz.K_bch * sizeof(int32_t)));
    }
    // the payload sources of the reference that are files or constants (the modules themselves are StreamPU's): Source_user = text, `n_frames`, `K`, then the bits, cycled through;
    // Source_user_binary = any file, eight payload bits per byte, least significant first, started over at its end (--src-no-loop: last frame zero-padded, then done); Source_AZCW = zeros
    std::vector<int32_t> pattern;      // USER: [n_pat][K_bch]
    std::vector<unsigned char> blob;   // USER_BIN
    size_t n_pat = 0, src_pos = 0;
    bool src_done = false;
    if (src_type == "USER") {
        std::ifstream f(src_path);
        long long np = 0, k = 0;
        if (!(f >> np >> k) || np < 1 || k != sz.K_bch) { std::fprintf(stderr, "'%s' is not a source pattern file of %d-bit frames\n", src_path.c_str(), (int)sz.K_bch); return 2; }
        pattern.resize((size_t)np * sz.K_bch);
        for (auto &b : pattern) { int v; if (!(f >> v) || (v != 0 && v != 1)) { std::fprintf(stderr, "'%s' is truncated or holds something else than bits\n", src_path.c_str()); return 2; } b = v; }
        n_pat = (size_t)np;
    } else if (src_type == "USER_BIN") {
        std::ifstream f(src_path, std::ios::binary);
        blob.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        if (blob.empty() || sz.K_bch % 8) { std::fprintf(stderr, "'%s' is empty or unreadable\n", src_path.c_str()); return 2; }
    }
    std::vector<int32_t> h_info(src_type == "RAND" ? 0 : (size_t)F * sz.K_bch, 0);
    auto generate = [&]() -> bool {      // the next F frames into h_info; false: the source is done
        if (src_type == "USER") {
            for (int f = 0; f < F; f++) { std::memcpy(&h_info[(size_t)f * sz.K_bch], &pattern[(src_pos % n_pat) * sz.K_bch], (size_t)sz.K_bch * sizeof(int32_t)); src_pos++; }
        } else if (src_type == "USER_BIN") {
            if (src_done) return false;
            const size_t nb = (size_t)F * sz.K_bch / 8;
            for (size_t i = 0; i < nb; i++) {
                unsigned char byte = 0;
                if (!src_no_loop) { byte = blob[src_pos % blob.size()]; src_pos++; }
                else if (src_pos < blob.size()) byte = blob[src_pos++];
                for (int b = 0; b < 8; b++) h_info[i * 8 + b] = (byte >> b) & 1;
            }
            if (src_no_loop && src_pos >= blob.size()) src_done = true;
        }
        return true;                     // (AZCW: h_info stays zero)
    };
    const bool chief = rank == 0;

    if (chief) std::printf("# * DVB-S2 (HIP) ------------------------------------\n#    ** Modulation and coding = %s\n#    ** LDPC implem           = %s\n"
                "#    ** LDPC n iterations     = %d\n#    ** Frames per batch (-F)  = %d\n#    ** Clones per process     = %d\n#    ** Processes (1 per GPU)  = %d\n", modcod.c_str(), implem.c_str(), n_ite, F, n_clones, world);
    if (chief) std::printf("# ----------|----------||----------|----------|----------|----------|----------||----------|----------\n"
                "#     Es/N0 |    Eb/N0 ||      FRA |       BE |       FE |      BER |      FER ||  SIM_THR |    ET/RT\n"
                "#      (dB) |     (dB) ||          |          |          |          |          ||   (Mb/s) | (hhmmss)\n"
                "# ----------|----------||----------|----------|----------|----------|----------||----------|----------\n");
    const double R = (double)sz.K_bch / sz.N_ldpc;                       // main.cpp:142
    unsigned long long batch = (unsigned long long)rank << 40;            // disjoint Philox key ranges per rank
    for (double ebn0 = ebn0_min; ebn0 < ebn0_max - 1e-9; ebn0 += step) {
        const double esn0 = ebn0 + 10.0 * std::log10(R * sz.bps);         // main.cpp:143-146
        const float sigma = (float)std::sqrt(1.0 / (2.0 * std::pow(10.0, esn0 / 10.0)));
        std::vector<float> sig(F, sigma);
        for (auto &k : cl) { h = k.h; CHK(dvbs2hip_memcpy_h2d(h, k.d_sig, sig.data(), sig.size() * sizeof(float))); CHK(dvbs2hip_monitor_reset(h)); k.c[0] = k.c[1] = k.c[2] = 0; k.busy = false; }
        src_pos = 0; src_done = false;                                          // the source starts over at every noise point
        uint64_t c[3] = {0, 0, 0};
        // a clone's counters run on the device from the reset above; the sum over the clones of what each one last reported is what the stopping rule sees
        auto collect = [&](Clone &k) -> int {
            h = k.h;
            CHK(dvbs2hip_monitor_reduce(h, k.c));                             // Monitor_reduction::is_done_all: every rank sees the same sum
            k.busy = false;
            for (int i = 0; i < 3; i++) { c[i] = 0; for (auto &o : cl) c[i] += o.c[i]; }
            return 0;
        };
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t turn = 0;; turn++) {
            Clone &k = cl[turn % cl.size()];
            if (k.busy) { const int rc = collect(k); if (rc) return rc; }
            if (c[2] >= (uint64_t)max_fe || (long long)c[0] >= max_frames) break;     // Monitor_BFER: stop at max_fe (DVBS2.cpp:136)
            h = k.h;
            if (src_type != "RAND") {
                if (!generate()) break;                                         // --src-type USER_BIN --src-no-loop: the file has been sent
                CHK(dvbs2hip_memcpy_h2d(h, k.d_info, h_info.data(), h_info.size() * sizeof(int32_t)));      // (returns when the copy is done: h_info is free for the next clone)
            }
            CHK(dvbs2hip_tx_bb_dev(h, (const int32_t *)k.d_info, (batch++ << 8), (const float *)k.d_sig, (int32_t *)k.d_sent, (float *)k.d_pl, F));
            CHK(dvbs2hip_rx_bb_dev(h, (const float *)k.d_pl, est == "PERFECT" ? (const float *)k.d_sig : nullptr, (int32_t *)k.d_got, nullptr, nullptr, F));
            CHK(dvbs2hip_monitor_check_errors_dev(h, (const int32_t *)k.d_sent, (const int32_t *)k.d_got, F));
            k.busy = true;
        }
        for (auto &k : cl) if (k.busy) { const int rc = collect(k); if (rc) return rc; }      // the batches still in flight count (the reference's threads finish theirs)
        const double et = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const int hh = (int)(et / 3600), mm = (int)(et / 60) % 60, ss = (int)et % 60;
        if (chief) std::printf("  %9.2f | %8.2f || %8llu | %8llu | %8llu | %8.2e | %8.2e || %8.3f | %02dh%02d'%02d\n", esn0, ebn0, (unsigned long long)c[0],
                    (unsigned long long)c[1], (unsigned long long)c[2], (double)c[1] / ((double)c[0] * sz.K_bch), (double)c[2] / (double)c[0],
                    (double)c[0] * sz.K_bch / et / 1e6, hh, mm, ss);
        std::fflush(stdout);
    }
    if (chief) std::printf("# End of the simulation\n");
    if (sim_stats && chief) {      // device time per kernel group, summed over the clones of this process (hipEvents around every launch: include/dvbs2hip.h, "measurement")
        static const char *names[DVBS2HIP_K_COUNT] = {"LDPC decoder", "BCH decoder", "demodulator", "filters", "front end (descramble + estimate + demodulate)", "other (TX mirror, synchronizers, monitor)"};
        double ms[DVBS2HIP_K_COUNT] = {0}, all = 0; long long n[DVBS2HIP_K_COUNT] = {0};
        for (auto &k : cl)
            for (int g = 0; g < DVBS2HIP_K_COUNT; g++) { double m1 = 0; int64_t n1 = 0; h = k.h; CHK(dvbs2hip_timing_get(h, g, &m1, &n1)); ms[g] += m1; n[g] += (long long)n1; all += m1; }
        std::printf("# -------------------------------------------------||------------||------------||---------\n#                                     Kernel group ||   launches || device (ms) ||    share\n"
                    "# -------------------------------------------------||------------||------------||---------\n");
        for (int g = 0; g < DVBS2HIP_K_COUNT; g++) if (n[g]) std::printf("# %48s || %10lld || %10.2f || %6.1f %%\n", names[g], n[g], ms[g], 100.0 * ms[g] / (all > 0 ? all : 1));
    }
    for (auto &k : cl) { if (k.d_info) dvbs2hip_free(k.h, k.d_info); dvbs2hip_free(k.h, k.d_pl); dvbs2hip_free(k.h, k.d_sent); dvbs2hip_free(k.h, k.d_got); dvbs2hip_free(k.h, k.d_sig); dvbs2hip_destroy(k.h); }
    return 0;
}
