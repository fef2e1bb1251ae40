// dvbs2_rx_bb -- the RX graphs of the reference wired with the HIP modules (host/dvbs2hip_modules.hpp), fed from raw IQ
// files in the reference's Radio_user_binary format (native-endian interleaved re/im float32:
// src/common/Module/Radio/Radio_user_binary/Radio_user_binary.cpp:55-100).
//
//  (1) the RX half of dvbs2_tx_rx_bb (src/mains/TX_RX_BB/main.cpp:83-94), 2 * pl_frame values per frame:
//   dvbs2_rx_bb --mod-cod QPSK-S_8/9 -F 8 --dec-implem NMS --dec-ite 10 --in pl_frames.f32 --out info_bits.i32 [--src sent_bits.i32] [--frame-sync]
//      --frame-sync puts Synchronizer_frame_hip::synchronize in front, as src/mains/RX/main.cpp does; the input may then start
//      anywhere in a frame.  The SAME batch runs twice -- task by task through the sockets of the reference graph, and through
//      the fused Receiver_BB_hip task -- and the program fails if they differ.
//
//  (2) --matched-filter: the dvbs2_rx graph from the matched filter to the monitor (src/mains/RX/main_sched.cpp:199-223,
//      probes :244-247), 2 * pl_frame * osf values per frame (osf = 2 samples per symbol):
//   dvbs2_rx_bb --matched-filter --mod-cod QPSK-S_8/9 -F 4 --in shaped_stream.f32 --src sent_bits.i32 --src-delay 1 --mon-skip 1 --out info_bits.i32
//      The binding lines between the modules built here are the reference's, character for character (marked "main_sched.cpp:NNN").
//      The sample-serial loops in between are out of scope and are played by stand-ins defined below:
//      sync_coarse_f = identity (no frequency offset), sync_timing = decimation by osf at the even phase (perfect timing),
//      no AGC; the source is a file of the sent payloads delayed by --src-delay frames (the frame synchronizer's latency;
//      what Filter_buffered_delay does in the TX_RX mains).
//
// With --src the monitor runs (check_errors in (1), check_errors2 with its BE / FE / BER / FER sockets read by probe stand-ins
// in (2)) and FRA / BE / FE are printed like the reference's terminal.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include "dvbs2hip_modules.hpp"

using namespace aff3ct;

// ---- stand-ins for StreamPU / dvbs2 modules that are outside the hot path: just enough surface for the binding lines
namespace spu { namespace module {
namespace src { enum class tsk : size_t { generate }; namespace sck { enum class generate : size_t { out_data, status }; } }
namespace prb { enum class tsk : size_t { probe }; namespace sck { enum class probe : size_t { in, status }; } }
class Source_file : public Module {          // Source_user_binary: payloads from a file, delayed by `delay` frames
public:
    Source_file(int K, int n_frames_, const std::string &path, int delay) : K(K), in(path, std::ios::binary), fifo((size_t)delay * K, 0)
    {
        if (!in) throw tools::runtime_error(__FILE__, __LINE__, __func__, "cannot open " + path);
        n_frames = (size_t)n_frames_;
        auto &t = create_task("generate");
        auto so = create_socket_out<int>(t, "out_data", K);
        create_codelet(t, [so](Module &m, runtime::Task &tk, size_t) -> int {
            auto &s = static_cast<Source_file &>(m);
            int *o = tk[so].get_dataptr<int>();
            const size_t n = (size_t)s.K * s.n_frames;
            std::vector<int> fresh(n, 0);
            s.in.read(reinterpret_cast<char *>(fresh.data()), n * sizeof(int));
            s.fifo.insert(s.fifo.end(), fresh.begin(), fresh.end());
            std::copy(s.fifo.begin(), s.fifo.begin() + n, o);
            s.fifo.erase(s.fifo.begin(), s.fifo.begin() + n);
            return 0;
        });
    }
    runtime::Socket &operator[](src::sck::generate s) { return (*tasks[0])[(size_t)s]; }
    runtime::Task &operator[](src::tsk) { return *tasks[0]; }
private:
    int K; std::ifstream in; std::vector<int> fifo;
};
template <typename T> class Probe_value : public Module {     // keeps the last value of the socket it is bound to
public:
    Probe_value(int size, const std::string &col, int n_frames_) : col(col)
    {
        n_frames = (size_t)n_frames_;
        auto &t = create_task("probe");
        auto si = create_socket_in<T>(t, "in", size);
        create_codelet(t, [si](Module &m, runtime::Task &tk, size_t) -> int {
            auto &p = static_cast<Probe_value &>(m);
            p.last = tk[si].template get_dataptr<const T>()[p.n_frames - 1];
            return 0;
        });
    }
    runtime::Socket &operator[](prb::sck::probe s) { return (*tasks[0])[(size_t)s]; }
    runtime::Task &operator[](prb::tsk) { return *tasks[0]; }
    T last = T();
    std::string col;
};
}}  // namespace spu::module
namespace aff3ct { namespace module {
class Sync_timing_perfect : public spu::module::Module {        // stand-in for Synchronizer_timing (Gardner, out of scope): even phase of osf = 2
public:
    Sync_timing_perfect(int N_in, int osf, int n_frames_)
    {
        n_frames = (size_t)n_frames_;
        auto &t = create_task("extract");
        auto sx = create_socket_in<float>(t, "X_N1", N_in);
        auto sy = create_socket_out<float>(t, "Y_N2", N_in / osf);
        create_codelet(t, [sx, sy, osf](spu::module::Module &, spu::runtime::Task &tk, size_t) -> int {
            const float *x = tk[sx].get_dataptr<const float>();
            float *y = tk[sy].get_dataptr<float>();
            const size_t n = tk[sy].get_n_elmts() / 2;
            for (size_t i = 0; i < n; i++) { y[2 * i] = x[2 * i * osf]; y[2 * i + 1] = x[2 * i * osf + 1]; }
            return 0;
        });
    }
    spu::runtime::Socket &in() { return (*tasks[0])[0]; }
    spu::runtime::Socket &out() { return (*tasks[0])[1]; }
    spu::runtime::Task &operator()() { return *tasks[0]; }
};
}}  // namespace aff3ct::module

static int run_matched_filter_graph(const std::string &modcod, int F, int n_ite, float alpha, const std::string &implem, const std::string &in_path,
                                    const std::string &out_path, const std::string &src_path, int src_delay, int mon_skip, float coarse_freq)
{
    using namespace module;
    const int osf = 2;
    auto ctx = std::make_shared<Context>(modcod, F, n_ite, alpha, true, 0, implem);
    const int N_pl = 2 * ctx->sz.pl_frame_sym;
    std::vector<float> rx_samples((size_t)F * N_pl * osf);                       // what Radio::receive hands over
    // modules: unique_ptr + the reference's variable names, so that the binding lines below are the reference's own
    std::unique_ptr<Multiplier_AGC_hip>          front_agc    (new Multiplier_AGC_hip(ctx, N_pl * osf, 1.f / (float)osf));                // DVBS2.cpp:660-664 (build_channel_agc)
    std::unique_ptr<Multiplier_AGC_hip>          mult_agc     (new Multiplier_AGC_hip(ctx, N_pl, 1.f));                                  // DVBS2.cpp:653-657 (build_agc_shift)
    std::unique_ptr<Synchronizer_freq_coarse_hip<>> sync_coarse_f(new Synchronizer_freq_coarse_hip<>(ctx, N_pl * osf));                  // the frequency shift of the transmission phase
    sync_coarse_f->set_curr_freq(coarse_freq);                                                                                          // (what the reference's loop would have settled on: --coarse-freq)
    std::unique_ptr<Filter_FIR_hip>              matched_flt  (new Filter_FIR_hip(ctx, N_pl * osf));
    std::unique_ptr<Sync_timing_perfect>         sync_timing  (new Sync_timing_perfect(N_pl * osf, osf, F));
    std::unique_ptr<Synchronizer_frame_hip<>>    sync_frame   (new Synchronizer_frame_hip<>(ctx));
    std::unique_ptr<Scrambler_PL_hip>            pl_scrambler (new Scrambler_PL_hip(ctx));
    std::unique_ptr<Synchronizer_freq_fine_hip<>> sync_fine_lr(new Synchronizer_freq_fine_hip<>(ctx, true));
    std::unique_ptr<Synchronizer_freq_fine_hip<>> sync_fine_pf(new Synchronizer_freq_fine_hip<>(ctx, false));
    std::unique_ptr<Framer_hip>                  framer       (new Framer_hip(ctx));
    std::unique_ptr<Estimator_hip>               estimator    (new Estimator_hip(ctx));
    std::unique_ptr<Modem_hip<>>                 modem        (new Modem_hip<>(ctx));
    std::unique_ptr<Interleaver_hip>             itl_rx       (new Interleaver_hip(ctx));
    std::unique_ptr<Decoder_LDPC_hip<>>          LDPC_decoder (new Decoder_LDPC_hip<>(ctx));
    std::unique_ptr<Decoder_BCH_hip<>>           BCH_decoder  (new Decoder_BCH_hip<>(ctx));
    std::unique_ptr<Scrambler_BB_hip<>>          bb_scrambler (new Scrambler_BB_hip<>(ctx));
    std::unique_ptr<Monitor_BFER_hip<>>          monitor      (new Monitor_BFER_hip<>(ctx));
    std::unique_ptr<spu::module::Source_file>    source       (new spu::module::Source_file(ctx->sz.K_bch, F, src_path, src_delay));
    spu::module::Probe_value<int32_t> prb_bfer_be(1, "BE", F);                   // main_sched.cpp:153-156
    spu::module::Probe_value<int32_t> prb_bfer_fe(1, "FE", F);
    spu::module::Probe_value<float> prb_bfer_ber(1, "BER", F);
    spu::module::Probe_value<float> prb_bfer_fer(1, "FER", F);

    (*front_agc    )[             mlt::sck::imultiply    ::X_N     ] = rx_samples;                                                                 // main_sched.cpp:197 (producer = the radio's socket)
    (*sync_coarse_f)[             sfc::sck::synchronize  ::X_N1    ] = (*front_agc    )[             mlt::sck::imultiply    ::Z_N     ];   // main_sched.cpp:198
    (*matched_flt  )[             flt::sck::filter1      ::X_N1    ] = (*sync_coarse_f)[             sfc::sck::synchronize  ::Y_N2    ];   // main_sched.cpp:199
    (*matched_flt  )[             flt::sck::filter2      ::X_N1    ] = (*sync_coarse_f)[             sfc::sck::synchronize  ::Y_N2    ];   // main_sched.cpp:200
    (*matched_flt  )[             flt::sck::filter2      ::Y_N2h   ] = (*matched_flt  )[             flt::sck::filter1      ::Y_N2    ];   // main_sched.cpp:201
    sync_timing->in() = (*matched_flt  )[             flt::sck::filter2      ::Y_N2    ];                                                // stand-in for :202-204 (Gardner)
    (*mult_agc     )[             mlt::sck::imultiply    ::X_N     ] = sync_timing->out();                                              // main_sched.cpp:205, producer = the stand-in
    (*sync_frame   )[             sfm::sck::synchronize1 ::X_N1    ] = (*mult_agc     )[             mlt::sck::imultiply    ::Z_N     ];   // main_sched.cpp:206
    (*sync_frame   )[             sfm::sck::synchronize2 ::X_N1    ] = (*mult_agc     )[             mlt::sck::imultiply    ::Z_N     ];   // main_sched.cpp:207
    (*sync_frame   )[             sfm::sck::synchronize2 ::cor_SOF ] = (*sync_frame   )[             sfm::sck::synchronize1 ::cor_SOF ];   // main_sched.cpp:208
    (*sync_frame   )[             sfm::sck::synchronize2 ::cor_PLSC] = (*sync_frame   )[             sfm::sck::synchronize1 ::cor_PLSC];   // main_sched.cpp:209
    (*pl_scrambler )[             scr::sck::descramble   ::Y_N1    ] = (*sync_frame   )[             sfm::sck::synchronize2 ::Y_N2    ];   // main_sched.cpp:210
    (*sync_fine_lr )[             sff::sck::synchronize  ::X_N1    ] = (*pl_scrambler )[             scr::sck::descramble   ::Y_N2    ];   // main_sched.cpp:211
    (*sync_fine_pf )[             sff::sck::synchronize  ::X_N1    ] = (*sync_fine_lr )[             sff::sck::synchronize  ::Y_N2    ];   // main_sched.cpp:212
    (*framer       )[             frm::sck::remove_plh   ::Y_N1    ] = (*sync_fine_pf )[             sff::sck::synchronize  ::Y_N2    ];   // main_sched.cpp:213
    (*estimator    )[             est::sck::estimate     ::X_N     ] = (*framer       )[             frm::sck::remove_plh   ::Y_N2    ];   // main_sched.cpp:214
    (*modem        )[             mdm::sck::demodulate   ::CP      ] = (*estimator    )[             est::sck::estimate     ::SIG     ];   // main_sched.cpp:215
    (*modem        )[             mdm::sck::demodulate   ::Y_N1    ] = (*framer       )[             frm::sck::remove_plh   ::Y_N2    ];   // main_sched.cpp:216
    (*itl_rx       )[             itl::sck::deinterleave ::itl     ] = (*modem        )[             mdm::sck::demodulate   ::Y_N2    ];   // main_sched.cpp:217
    (*LDPC_decoder )[             dec::sck::decode_siho  ::Y_N     ] = (*itl_rx       )[             itl::sck::deinterleave ::nat     ];   // main_sched.cpp:218
    (*BCH_decoder  )[             dec::sck::decode_hiho  ::Y_N     ] = (*LDPC_decoder )[             dec::sck::decode_siho  ::V_K     ];   // main_sched.cpp:219
    (*bb_scrambler )[             scr::sck::descramble   ::Y_N1    ] = (*BCH_decoder  )[             dec::sck::decode_hiho  ::V_K     ];   // main_sched.cpp:220
    (*monitor      )[             mnt::sck::check_errors2::U       ] = (*source       )[spu::module::src::sck::generate     ::out_data];   // main_sched.cpp:221
    (*monitor      )[             mnt::sck::check_errors2::V       ] = (*bb_scrambler )[             scr::sck::descramble   ::Y_N2    ];   // main_sched.cpp:222
    prb_bfer_be     [spu::module::prb::sck::probe        ::in      ] = (*monitor      )[             mnt::sck::check_errors2::BE      ];   // main_sched.cpp:244
    prb_bfer_fe     [spu::module::prb::sck::probe        ::in      ] = (*monitor      )[             mnt::sck::check_errors2::FE      ];   // main_sched.cpp:245
    prb_bfer_ber    [spu::module::prb::sck::probe        ::in      ] = (*monitor      )[             mnt::sck::check_errors2::BER     ];   // main_sched.cpp:246
    prb_bfer_fer    [spu::module::prb::sck::probe        ::in      ] = (*monitor      )[             mnt::sck::check_errors2::FER     ];   // main_sched.cpp:247

    spu::runtime::Sequence seq({&(*source)[spu::module::src::tsk::generate], &(*front_agc)(), &(*sync_coarse_f)(), &(*matched_flt)[flt::tsk::filter1], &(*matched_flt)[flt::tsk::filter2],
                                &(*sync_timing)(), &(*mult_agc)(), &(*sync_frame)[sfm::tsk::synchronize1], &(*sync_frame)[sfm::tsk::synchronize2], &(*pl_scrambler)(),
                                &(*sync_fine_lr)(), &(*sync_fine_pf)(), &(*framer)(), &(*estimator)(), &(*modem)(), &(*itl_rx)(), &(*LDPC_decoder)(),
                                &(*BCH_decoder)(), &(*bb_scrambler)(), &(*monitor)[mnt::tsk::check_errors2], &prb_bfer_be[spu::module::prb::tsk::probe],
                                &prb_bfer_fe[spu::module::prb::tsk::probe], &prb_bfer_ber[spu::module::prb::tsk::probe], &prb_bfer_fer[spu::module::prb::tsk::probe]});

    std::ifstream in(in_path, std::ios::binary);
    if (!in) throw spu::tools::runtime_error(__FILE__, __LINE__, __func__, "cannot open " + in_path);
    std::ofstream out;
    if (!out_path.empty()) out.open(out_path, std::ios::binary);
    size_t batches = 0;
    while (in.read(reinterpret_cast<char *>(rx_samples.data()), rx_samples.size() * sizeof(float))) {
        seq.exec_step();
        if ((int)batches < mon_skip) monitor->reset();       // the synchronizers are still locking: keep these batches out of the statistics (the reference's waiting and learning phases)
        // The fine frequency estimate of ONE frame is noisy (sigma ~2.5e-4 cycles per symbol at 16 dB against the +-3.4e-4 the pilot-aided stage can take back): it lives on its memory of
        // alpha = 0.999.  In the reference the fine synchronizers do not run before the frame synchronizer has locked -- its learning phases 1 and 2 end at sync_frame
        // (main_sched.cpp:454,535) -- and then settle for a learning phase 3 (:597-630) before anything is counted.  Here: the first half of --mon-skip is phases 1 and 2 (what the
        // estimate saw of the unaligned frames is dropped), the second half is phase 3.
        if ((int)batches < (mon_skip + 1) / 2) sync_fine_lr->reset();
        if (out.is_open())
            out.write(reinterpret_cast<const char *>((*bb_scrambler)[scr::sck::descramble::Y_N2].get_dataptr<int>()), (size_t)F * ctx->sz.K_bch * sizeof(int));
        batches++;
    }
    uint64_t fra = 0, be = 0, fe = 0;
    monitor->get(fra, be, fe);
    std::printf("# %s F=%d %s ite=%d matched-filter graph | batches %zu | FRA %llu BE %llu FE %llu\n", modcod.c_str(), F, implem.c_str(), n_ite, batches,
                (unsigned long long)fra, (unsigned long long)be, (unsigned long long)fe);
    std::printf("# probes | BE %d FE %d BER %.3e FER %.3e | DEL %d FLG %d\n", prb_bfer_be.last, prb_bfer_fe.last, (double)prb_bfer_ber.last, (double)prb_bfer_fer.last,
                (*sync_frame)[sfm::sck::synchronize2::DEL].get_dataptr<int>()[F - 1], (*sync_frame)[sfm::sck::synchronize2::FLG].get_dataptr<int>()[F - 1]);
    return 0;
}

int main(int argc, char **argv)
{
    std::string modcod = "QPSK-S_8/9", in_path, out_path, src_path, implem = "SPA";     // DVBS2.cpp:135-138: SPA, 50 iterations
    int F = 1, n_ite = 50, src_delay = 0, mon_skip = 0;
    float coarse_freq = 0.f;
    float alpha = 1.0f;
    bool frame_sync = false, matched = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << "\n"; exit(2); } return argv[++i]; };
        if (a == "--mod-cod") modcod = next();
        else if (a == "-F" || a == "--sim-inter-fra") F = std::stoi(next());
        else if (a == "--dec-ite") n_ite = std::stoi(next());
        else if (a == "--dec-implem") implem = next();
        else if (a == "--dec-alpha") alpha = std::stof(next());
        else if (a == "--in") in_path = next();
        else if (a == "--out") out_path = next();
        else if (a == "--src") src_path = next();
        else if (a == "--src-delay") src_delay = std::stoi(next());
        else if (a == "--mon-skip") mon_skip = std::stoi(next());
        else if (a == "--coarse-freq") coarse_freq = std::stof(next());      // normalized frequency offset of the received samples (cycles per sample): the coarse synchronizer's frozen estimate
        else if (a == "--frame-sync") frame_sync = true;
        else if (a == "--matched-filter") matched = true;
        else { std::cerr << "unknown argument " << a << "\n"; return 2; }
    }
    try {
        if (matched) return run_matched_filter_graph(modcod, F, n_ite, alpha, implem, in_path, out_path, src_path, src_delay, mon_skip, coarse_freq);
        auto ctx = std::make_shared<module::Context>(modcod, F, n_ite, alpha, true, 0, implem);
        ctx->pin_sockets = true;            // the sockets below live until the modules go: pin them for overlapped PCIe copies
        module::Scrambler_PL_hip pl_scrambler(ctx);
        module::Framer_hip framer(ctx);
        module::Estimator_hip estimator(ctx);
        module::Modem_hip<> modem(ctx);
        module::Interleaver_hip itl_rx(ctx);
        module::Decoder_LDPC_hip<> LDPC_decoder(ctx);
        module::Decoder_BCH_hip<> BCH_decoder(ctx);
        module::Scrambler_BB_hip<> bb_scrambler(ctx);
        module::Monitor_BFER_hip<> monitor(ctx);
        module::Receiver_BB_hip<> receiver(ctx);

        std::vector<float> pl_frames((size_t)F * 2 * ctx->sz.pl_frame_sym);
        std::vector<int> sent((size_t)F * ctx->sz.K_bch, 0);

        // socket binding, as TX_RX_BB/main.cpp:83-94 (RX side)
        using namespace module;
        pl_scrambler[scr::sck::descramble::Y_N1] = pl_frames;
        framer      [frm::sck::remove_plh::Y_N1] = pl_scrambler[scr::sck::descramble::Y_N2];
        estimator   [est::sck::estimate::X_N   ] = framer      [frm::sck::remove_plh::Y_N2];
        modem       [mdm::sck::demodulate::CP  ] = estimator   [est::sck::estimate::SIG   ];
        modem       [mdm::sck::demodulate::Y_N1] = framer      [frm::sck::remove_plh::Y_N2];
        itl_rx      [itl::sck::deinterleave::itl] = modem      [mdm::sck::demodulate::Y_N2];
        LDPC_decoder[dec::sck::decode_siho::Y_N] = itl_rx      [itl::sck::deinterleave::nat];
        BCH_decoder [dec::sck::decode_hiho::Y_N] = LDPC_decoder[dec::sck::decode_siho::V_K];
        bb_scrambler[scr::sck::descramble::Y_N1] = BCH_decoder [dec::sck::decode_hiho::V_K];
        monitor     [mnt::sck::check_errors::U ] = sent;
        monitor     [mnt::sck::check_errors::V ] = bb_scrambler[scr::sck::descramble::Y_N2];
        receiver    [rcv::sck::receive::Y_N1   ] = pl_frames;

        std::vector<spu::runtime::Task *> order = {&pl_scrambler(), &framer(), &estimator(), &modem(), &itl_rx(),
                                                   &LDPC_decoder(), &BCH_decoder(), &bb_scrambler()};
        std::unique_ptr<module::Synchronizer_frame_hip<>> sync_frame;
        if (frame_sync) {
            sync_frame.reset(new module::Synchronizer_frame_hip<>(ctx));
            (*sync_frame)[sfm::sck::synchronize::X_N1] = pl_frames;
            pl_scrambler[scr::sck::descramble::Y_N1] = (*sync_frame)[sfm::sck::synchronize::Y_N2];
            receiver    [rcv::sck::receive::Y_N1   ] = (*sync_frame)[sfm::sck::synchronize::Y_N2];
            order.insert(order.begin(), &(*sync_frame)[sfm::tsk::synchronize]);
        }
        if (!src_path.empty()) order.push_back(&monitor[mnt::tsk::check_errors]);
        spu::runtime::Sequence seq(order);

        std::ifstream in(in_path, std::ios::binary);
        if (!in) throw spu::tools::runtime_error(__FILE__, __LINE__, __func__, "cannot open " + in_path);
        std::ifstream src;
        if (!src_path.empty()) { src.open(src_path, std::ios::binary); if (!src) throw spu::tools::runtime_error(__FILE__, __LINE__, __func__, "cannot open " + src_path); }
        std::ofstream out;
        if (!out_path.empty()) out.open(out_path, std::ios::binary);

        size_t batches = 0, mismatch = 0;
        while (in.read(reinterpret_cast<char *>(pl_frames.data()), pl_frames.size() * sizeof(float))) {
            if (src.is_open()) src.read(reinterpret_cast<char *>(sent.data()), sent.size() * sizeof(int));
            seq.exec_step();            // the reference's graph, one task per codelet
            receiver().exec();          // the fused device-resident chain
            const int *a = bb_scrambler[scr::sck::descramble::Y_N2].get_dataptr<int>();
            const int *b = receiver[rcv::sck::receive::V_K].get_dataptr<int>();
            for (size_t i = 0; i < sent.size(); i++) mismatch += a[i] != b[i];
            if (out.is_open()) out.write(reinterpret_cast<const char *>(b), sent.size() * sizeof(int));
            batches++;
        }
        uint64_t fra = 0, be = 0, fe = 0;
        monitor.get(fra, be, fe);
        std::printf("# %s F=%d %s ite=%d | batches %zu | task-graph vs fused mismatches %zu | FRA %llu BE %llu FE %llu\n", modcod.c_str(), F, implem.c_str(),
                    n_ite, batches, mismatch, (unsigned long long)fra, (unsigned long long)be, (unsigned long long)fe);
        if (sync_frame)
            std::printf("# frame-sync | DEL %d FLG %d TRI %.3f\n", (*sync_frame)[sfm::sck::synchronize::DEL].get_dataptr<int>()[F - 1],
                        (*sync_frame)[sfm::sck::synchronize::FLG].get_dataptr<int>()[F - 1], (double)(*sync_frame)[sfm::sck::synchronize::TRI].get_dataptr<float>()[F - 1]);
        return mismatch ? 1 : 0;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 3;
    }
}
