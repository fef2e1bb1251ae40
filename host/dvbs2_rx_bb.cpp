// dvbs2_rx_bb -- the RX half of the reference's dvbs2_tx_rx_bb (src/mains/TX_RX_BB/main.cpp:83-94)
// wired with the HIP modules, fed from a raw IQ file in the reference's Radio_user_binary format
// (native-endian interleaved re/im float32, 2 * pl_frame values per frame:
//  src/common/Module/Radio/Radio_user_binary/Radio_user_binary.cpp:55-100).
//
//   dvbs2_rx_bb --mod-cod QPSK-S_8/9 -F 8 --dec-ite 10 --in pl_frames.f32 --out info_bits.i32 [--src sent_bits.i32] [--frame-sync]
//
// --frame-sync puts Synchronizer_frame_hip::synchronize in front, as the RX mains do
// (src/mains/RX/main.cpp: sync_frame Y_N2 -> pl_scrambler Y_N1): the input may then start anywhere in a frame.
//
// It runs the SAME batch twice -- task by task through the ten sockets of the reference graph,
// and through the fused Receiver_BB_hip task -- and fails if they differ; with --src it also
// drives Monitor_BFER_hip and prints FRA / BE / FE like the reference's terminal.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include "dvbs2hip_modules.hpp"

using namespace aff3ct;

int main(int argc, char **argv)
{
    std::string modcod = "QPSK-S_8/9", in_path, out_path, src_path;
    int F = 1, n_ite = 50;
    float alpha = 1.0f;
    bool frame_sync = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << "\n"; exit(2); } return argv[++i]; };
        if (a == "--mod-cod") modcod = next();
        else if (a == "-F" || a == "--sim-inter-fra") F = std::stoi(next());
        else if (a == "--dec-ite") n_ite = std::stoi(next());
        else if (a == "--dec-alpha") alpha = std::stof(next());
        else if (a == "--in") in_path = next();
        else if (a == "--out") out_path = next();
        else if (a == "--src") src_path = next();
        else if (a == "--frame-sync") frame_sync = true;
        else { std::cerr << "unknown argument " << a << "\n"; return 2; }
    }
    try {
        auto ctx = std::make_shared<module::Context>(modcod, F, n_ite, alpha, true);
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
        if (!src_path.empty()) order.push_back(&monitor());
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
        std::printf("# %s F=%d ite=%d | batches %zu | task-graph vs fused mismatches %zu | FRA %llu BE %llu FE %llu\n", modcod.c_str(), F,
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
