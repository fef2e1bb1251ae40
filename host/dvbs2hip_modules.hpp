// dvbs2hip_modules.hpp -- StreamPU-style modules whose codelets call libdvbs2hip.so (the C ABI
// of include/dvbs2hip.h).  Task and socket names are the reference's, so the socket graph of
// /root/reference src/mains/TX_RX_BB/main.cpp:83-94 binds unchanged, and so do the lines of src/mains/RX/main_sched.cpp:197-247
// that touch the modules built here (matched filter filter1 / filter2 with Y_N2h, frame synchronizer synchronize1 / 2, the two
// fine synchronizers, PL descrambler .. BB descrambler, monitor check_errors2 with its BE / FE / BER / FER sockets); host/
// dvbs2_rx_bb.cpp holds those lines literally and tests/test_host_cpp.py builds and runs them.  The sample-serial loops of
// dvbs2_rx (AGC, coarse frequency PLL, Gardner timing) are out of scope and have no module here.  Each module replaces the
// aff3ct / dvbs2 module named in its comment.
//
// One GPU = one Context (dvbs2hip handle) shared by the modules of a chain; -F (n_frames) is the
// grid width on the device, so ONE module instance per GPU replaces the 28-40 thread clones of
// the reference's decoder stage (main_bench.cpp:156-168).  Decoder success is data (CWD socket),
// errors are spu::tools exceptions.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "../include/dvbs2hip.h"
#include "spu_compat.hpp"

namespace aff3ct {
namespace module {

class Context {     // RAII owner of the dvbs2hip handle
public:
    // defaults = the reference's (DVBS2.cpp:135-138: --dec-implem SPA, --dec-ite 50); ldpc_implem: "SPA" | "NMS" | "MS"
    Context(const std::string &modcod, int n_frames, int ldpc_n_ite = 50, float ldpc_alpha = 1.0f, bool early_stop = true,
            int device = 0, const std::string &ldpc_implem = "SPA")
    {
        dvbs2hip_cfg cfg;
        if (dvbs2hip_cfg_from_modcod(modcod.c_str(), &cfg) != DVBS2HIP_OK)
            throw spu::tools::invalid_argument(__FILE__, __LINE__, __func__, dvbs2hip_last_error(nullptr));   // DVBS2.cpp:319
        if (ldpc_implem != "SPA" && ldpc_implem != "SPA_TANH" && ldpc_implem != "SPA_EXACT" && ldpc_implem != "NMS" && ldpc_implem != "MS")
            throw spu::tools::invalid_argument(__FILE__, __LINE__, __func__, "'ldpc_implem' has to be SPA, SPA_TANH, SPA_EXACT, NMS or MS");
        cfg.ldpc_implem = ldpc_implem == "SPA" ? DVBS2HIP_IMPLEM_SPA : ldpc_implem == "SPA_TANH" ? DVBS2HIP_IMPLEM_SPA_TANH : ldpc_implem == "SPA_EXACT" ? DVBS2HIP_IMPLEM_SPA_EXACT : ldpc_implem == "MS" ? DVBS2HIP_IMPLEM_MS : DVBS2HIP_IMPLEM_NMS;
        cfg.max_frames = n_frames; cfg.ldpc_n_ite = ldpc_n_ite; cfg.ldpc_alpha = ldpc_alpha;
        cfg.ldpc_early_stop = early_stop ? 1 : 0; cfg.device = device;
        const int rc = dvbs2hip_create(&cfg, &h);
        if (rc != DVBS2HIP_OK) raise(rc, dvbs2hip_last_error(nullptr), __LINE__, __func__);
        dvbs2hip_get_sizes(h, &sz);
        n_frames_ = n_frames;
    }
    ~Context() { dvbs2hip_destroy(h); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    void check(int rc, int line, const char *func) const { if (rc != DVBS2HIP_OK) raise(rc, dvbs2hip_last_error(h), line, func); }
    dvbs2hip_t *h = nullptr;
    dvbs2hip_sizes sz{};
    int n_frames() const { return n_frames_; }
    // Socket buffers live as long as their modules: with pin_sockets set, the decode_siho / receive codelets pin the
    // buffers they are handed (once: dvbs2hip_host_register ignores a buffer it knows) and the C ABI then overlaps the
    // PCIe copies with the kernels.  Leave it off if the graph is fed from short-lived buffers.
    bool pin_sockets = false;
    void pin(const void *p, size_t bytes) const { if (pin_sockets && p) (void)dvbs2hip_host_register(h, const_cast<void *>(p), bytes); }

private:
    int n_frames_ = 1;
    [[noreturn]] static void raise(int rc, const char *msg, int line, const char *func)
    {
        switch (rc) {
            case DVBS2HIP_EINVAL: throw spu::tools::invalid_argument(__FILE__, line, func, msg);
            case DVBS2HIP_ENOMEM: throw spu::tools::cannot_allocate(__FILE__, line, func, msg);
            case DVBS2HIP_EUNSUPPORTED: throw spu::tools::unimplemented_error(__FILE__, line, func, msg);
            default: throw spu::tools::runtime_error(__FILE__, line, func, msg);
        }
    }
};
#define DVBS2HIP_CHK(ctx, call) (ctx)->check((call), __LINE__, __func__)

namespace dec { enum class tsk : size_t { decode_siho = 0, decode_hiho = 0 };
                namespace sck { enum class decode_siho : size_t { Y_N, CWD, V_K, status }; enum class decode_hiho : size_t { Y_N, CWD, V_K, status }; } }
namespace mdm { namespace sck { enum class demodulate : size_t { CP, Y_N1, Y_N2, status }; } }
namespace itl { namespace sck { enum class deinterleave : size_t { itl, nat, status }; } }
// Filter.hpp:22-29
namespace flt { enum class tsk : size_t { filter, filter1, filter2, SIZE };
                namespace sck { enum class filter : size_t { X_N1, Y_N2, status }; enum class filter1 : size_t { X_N1, Y_N2, status };
                                enum class filter2 : size_t { X_N1, Y_N2h, Y_N2, status }; } }
namespace est { namespace sck { enum class estimate : size_t { X_N, SIG, Eb_N0, Es_N0, status }; } }
namespace scr { namespace sck { enum class descramble : size_t { Y_N1, Y_N2, status }; } }
namespace frm { namespace sck { enum class remove_plh : size_t { Y_N1, Y_N2, status }; } }
// Synchronizer_freq_coarse.hpp:14-19
namespace sfc { enum class tsk : size_t { synchronize, SIZE }; namespace sck { enum class synchronize : size_t { X_N1, FRQ, PHS, Y_N2, status }; } }
// Multiplier.hpp:16-26 (tasks imultiply, multiply; the gain stage has the first)
namespace mlt { enum class tsk : size_t { imultiply, multiply, SIZE }; namespace sck { enum class imultiply : size_t { X_N, Z_N, status }; } }
// aff3ct Monitor_BFER (absent submodule): tasks check_errors and check_errors2, the latter bound by the RX mains (main_sched.cpp:222-223,244-247)
namespace mnt { enum class tsk : size_t { check_errors, check_errors2, SIZE };
                namespace sck { enum class check_errors : size_t { U, V, status };
                                enum class check_errors2 : size_t { U, V, FRA, BE, FE, BER, FER, status }; } }
namespace rcv { namespace sck { enum class receive : size_t { Y_N1, V_K, CWD_LDPC, CWD_BCH, status }; } }
// Synchronizer_freq_fine.hpp:14-19
namespace sff { enum class tsk : size_t { synchronize, SIZE }; namespace sck { enum class synchronize : size_t { X_N1, FRQ, PHS, Y_N2, status }; } }
// Synchronizer_frame.hpp:15-24
namespace sfm { enum class tsk : size_t { synchronize, synchronize1, synchronize2, SIZE };
                namespace sck { enum class synchronize : size_t { X_N1, DEL, FLG, TRI, Y_N2, status };
                                enum class synchronize1 : size_t { X_N1, cor_SOF, cor_PLSC, status };
                                enum class synchronize2 : size_t { X_N1, cor_SOF, cor_PLSC, DEL, FLG, TRI, Y_N2, status }; } }

// common shape: one task, sockets created in enum order, codelet forwards raw pointers
class Module_hip : public spu::module::Stateful {
public:
    template <typename E> spu::runtime::Socket &operator[](E s) { return (*tasks[0])[(size_t)s]; }
    spu::runtime::Task &operator()(const std::string & = "") { return *tasks[0]; }
    void set_n_frames(size_t n) override
    {
        if ((int)n != ctx->n_frames())
            throw spu::tools::invalid_argument(__FILE__, __LINE__, __func__, "'n_frames' is fixed by the device context (the -F of the socket is the grid width)");
    }
protected:
    explicit Module_hip(std::shared_ptr<Context> c, const std::string &n) : ctx(std::move(c))
    {
        n_frames = (size_t)ctx->n_frames();
        set_name(n); set_short_name(n);
    }
    std::shared_ptr<Context> ctx;
    int F() const { return ctx->n_frames(); }
};

// replaces tools::Codec_LDPC<B,Q>::get_decoder_siho() (DVBS2.cpp:418-449; bound main.cpp:65,90-91)
template <typename B = int, typename Q = float>
class Decoder_LDPC_hip : public Module_hip {
public:
    explicit Decoder_LDPC_hip(std::shared_ptr<Context> c) : Module_hip(std::move(c), "Decoder_LDPC_hip")
    {
        static_assert(sizeof(B) == 4 && sizeof(Q) == 4, "B = int32, Q = float (SURVEY.md H6)");
        auto &t = create_task("decode_siho");
        auto sY = create_socket_in<Q>(t, "Y_N", ctx->sz.N_ldpc);
        auto sC = create_socket_out<int8_t>(t, "CWD", 1);
        auto sV = create_socket_out<B>(t, "V_K", ctx->sz.K_ldpc);
        create_codelet(t, [sY, sC, sV](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            auto &d = static_cast<Decoder_LDPC_hip &>(m);
            d.decode_siho(tk[sY].template get_dataptr<const Q>(), tk[sC].template get_dataptr<int8_t>(), tk[sV].template get_dataptr<B>());
            return 0;
        });
    }
    void decode_siho(const Q *Y_N, int8_t *CWD, B *V_K)
    {
        ctx->pin(Y_N, sizeof(Q) * (size_t)ctx->sz.N_ldpc * F()); ctx->pin(V_K, sizeof(B) * (size_t)ctx->sz.K_ldpc * F()); ctx->pin(CWD, (size_t)F());
        DVBS2HIP_CHK(ctx, dvbs2hip_ldpc_decode_siho(ctx->h, (const float *)Y_N, CWD, (int32_t *)V_K, F()));
    }
};

// replaces Decoder_BCH_DVBS2<B,R> (Decoder_BCH_DVBS2.cpp:28-40; built DVBS2.cpp:406-416)
template <typename B = int, typename R = float>
class Decoder_BCH_hip : public Module_hip {
public:
    explicit Decoder_BCH_hip(std::shared_ptr<Context> c) : Module_hip(std::move(c), "Decoder_BCH_hip")
    {
        auto &t = create_task("decode_hiho");
        auto sY = create_socket_in<B>(t, "Y_N", ctx->sz.K_ldpc);
        auto sC = create_socket_out<int8_t>(t, "CWD", 1);
        auto sV = create_socket_out<B>(t, "V_K", ctx->sz.K_bch);
        create_codelet(t, [sY, sC, sV](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Decoder_BCH_hip &>(m).decode_hiho(tk[sY].template get_dataptr<const B>(), tk[sC].template get_dataptr<int8_t>(), tk[sV].template get_dataptr<B>());
            return 0;
        });
    }
    void decode_hiho(const B *Y_N, int8_t *CWD, B *V_K)
    { DVBS2HIP_CHK(ctx, dvbs2hip_bch_decode_hiho(ctx->h, (const int32_t *)Y_N, CWD, (int32_t *)V_K, F())); }
    // the reference throws unimplemented_error for the soft / codeword variants (Decoder_BCH_DVBS2.cpp:42-61)
    void decode_siho(const R *, int8_t *, B *) { throw spu::tools::unimplemented_error(__FILE__, __LINE__, __func__); }
    void decode_hiho_cw(const B *, int8_t *, B *) { throw spu::tools::unimplemented_error(__FILE__, __LINE__, __func__); }
};

// replaces Modem_generic_fast<B,R,Q,max_star> demodulate (DVBS2.cpp:478-488; bound main.cpp:87-88)
template <typename Q = float>
class Modem_hip : public Module_hip {
public:
    explicit Modem_hip(std::shared_ptr<Context> c) : Module_hip(std::move(c), "Modem_hip")
    {
        auto &t = create_task("demodulate");
        auto sC = create_socket_in<float>(t, "CP", 1);
        auto s1 = create_socket_in<Q>(t, "Y_N1", 2 * ctx->sz.N_xfec_sym);
        auto s2 = create_socket_out<Q>(t, "Y_N2", ctx->sz.N_ldpc);
        create_codelet(t, [sC, s1, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Modem_hip &>(m).demodulate(tk[sC].template get_dataptr<const float>(), tk[s1].template get_dataptr<const Q>(), tk[s2].template get_dataptr<Q>());
            return 0;
        });
    }
    void demodulate(const float *CP, const Q *Y_N1, Q *Y_N2) { DVBS2HIP_CHK(ctx, dvbs2hip_demodulate(ctx->h, CP, Y_N1, Y_N2, F())); }
};

// replaces Interleaver<float,uint32_t>::deinterleave (DVBS2.cpp:451-476; bound main.cpp:56,89)
class Interleaver_hip : public Module_hip {
public:
    explicit Interleaver_hip(std::shared_ptr<Context> c) : Module_hip(std::move(c), "Interleaver_hip")
    {
        auto &t = create_task("deinterleave");
        auto s1 = create_socket_in<float>(t, "itl", ctx->sz.N_ldpc);
        auto s2 = create_socket_out<float>(t, "nat", ctx->sz.N_ldpc);
        create_codelet(t, [s1, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Interleaver_hip &>(m).deinterleave(tk[s1].template get_dataptr<const float>(), tk[s2].template get_dataptr<float>());
            return 0;
        });
    }
    void deinterleave(const float *itl, float *nat) { DVBS2HIP_CHK(ctx, dvbs2hip_deinterleave(ctx->h, itl, nat, F())); }
};

// replaces Filter_RRC_ccr_naive / Filter_FIR_ccr: tasks filter, filter1, filter2 (Filter.hpp:22-29, codelets Filter.hxx:56-96, bodies
// Filter_FIR_ccr.cpp:68-294); the RX mains bind filter1 / filter2 with the Y_N2h socket in between (main_sched.cpp:199-201)
class Filter_FIR_hip : public Module_hip {
public:
    Filter_FIR_hip(std::shared_ptr<Context> c, int N) : Module_hip(std::move(c), "Filter_FIR_hip"), N(N)
    {
        if (N <= 0 || N % 2) throw spu::tools::invalid_argument(__FILE__, __LINE__, __func__, "'N' has to be a positive even number of floats");
        {
            auto &t = create_task("filter");
            auto s1 = create_socket_in<float>(t, "X_N1", N);
            auto s2 = create_socket_out<float>(t, "Y_N2", N);
            create_codelet(t, [s1, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
                static_cast<Filter_FIR_hip &>(m).filter(tk[s1].template get_dataptr<const float>(), tk[s2].template get_dataptr<float>());
                return 0;
            });
        }
        {
            auto &t = create_task("filter1");
            auto s1 = create_socket_in<float>(t, "X_N1", N);
            auto s2 = create_socket_out<float>(t, "Y_N2", N);
            create_codelet(t, [s1, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
                static_cast<Filter_FIR_hip &>(m).filter1(tk[s1].template get_dataptr<const float>(), tk[s2].template get_dataptr<float>());
                return 0;
            });
        }
        {
            auto &t = create_task("filter2");
            auto s1 = create_socket_in<float>(t, "X_N1", N);
            auto sh = create_socket_in<float>(t, "Y_N2h", N);
            auto s2 = create_socket_out<float>(t, "Y_N2", N);
            create_codelet(t, [s1, sh, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
                static_cast<Filter_FIR_hip &>(m).filter2(tk[s1].template get_dataptr<const float>(), tk[sh].template get_dataptr<const float>(),
                                                         tk[s2].template get_dataptr<float>());
                return 0;
            });
        }
    }
    spu::runtime::Task &operator[](flt::tsk t) { return *tasks[(size_t)t]; }
    spu::runtime::Socket &operator[](flt::sck::filter s) { return (*tasks[(size_t)flt::tsk::filter])[(size_t)s]; }
    spu::runtime::Socket &operator[](flt::sck::filter1 s) { return (*tasks[(size_t)flt::tsk::filter1])[(size_t)s]; }
    spu::runtime::Socket &operator[](flt::sck::filter2 s) { return (*tasks[(size_t)flt::tsk::filter2])[(size_t)s]; }
    int get_N() const { return N; }
    void filter(const float *X_N1, float *Y_N2) { DVBS2HIP_CHK(ctx, dvbs2hip_filter(ctx->h, X_N1, Y_N2, N / 2, F())); }
    void filter1(const float *X_N1, float *Y_N2) { DVBS2HIP_CHK(ctx, dvbs2hip_filter1(ctx->h, X_N1, Y_N2, N / 2, F())); }
    void filter2(const float *X_N1, const float *Y_N2h, float *Y_N2) { DVBS2HIP_CHK(ctx, dvbs2hip_filter2(ctx->h, X_N1, Y_N2h, Y_N2, N / 2, F())); }
    void reset() { DVBS2HIP_CHK(ctx, dvbs2hip_filter_reset(ctx->h)); }
    // DVBS2HIP_FIR_AUTO (matrix cores for <= 81 taps) | DVBS2HIP_FIR_VALU | DVBS2HIP_FIR_MFMA
    void set_kernel(int kernel) { DVBS2HIP_CHK(ctx, dvbs2hip_set_filter_kernel(ctx->h, kernel)); }
private:
    int N;
};

// replaces Estimator_DVBS2<R>::estimate (Estimator_DVBS2.hxx:31-58, Estimator.hxx:103-118)
class Estimator_hip : public Module_hip {
public:
    explicit Estimator_hip(std::shared_ptr<Context> c) : Module_hip(std::move(c), "Estimator_hip")
    {
        auto &t = create_task("estimate");
        auto sX = create_socket_in<float>(t, "X_N", 2 * ctx->sz.N_xfec_sym);
        auto sS = create_socket_out<float>(t, "SIG", 1);
        auto sB = create_socket_out<float>(t, "Eb_N0", 1);
        auto sE = create_socket_out<float>(t, "Es_N0", 1);
        create_codelet(t, [sX, sS, sB, sE](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Estimator_hip &>(m).estimate(tk[sX].template get_dataptr<const float>(), tk[sS].template get_dataptr<float>(),
                                                     tk[sB].template get_dataptr<float>(), tk[sE].template get_dataptr<float>());
            return 0;
        });
    }
    void estimate(const float *X_N, float *SIG, float *Eb_N0, float *Es_N0)
    { DVBS2HIP_CHK(ctx, dvbs2hip_estimate(ctx->h, X_N, SIG, Eb_N0, Es_N0, F())); }
};

// replaces Scrambler_PL<D>::descramble (Scrambler_PL.hxx:61-78)
class Scrambler_PL_hip : public Module_hip {
public:
    explicit Scrambler_PL_hip(std::shared_ptr<Context> c) : Module_hip(std::move(c), "Scrambler_PL_hip")
    {
        auto &t = create_task("descramble");
        auto s1 = create_socket_in<float>(t, "Y_N1", 2 * ctx->sz.pl_frame_sym);
        auto s2 = create_socket_out<float>(t, "Y_N2", 2 * ctx->sz.pl_frame_sym);
        create_codelet(t, [s1, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Scrambler_PL_hip &>(m).descramble(tk[s1].template get_dataptr<const float>(), tk[s2].template get_dataptr<float>());
            return 0;
        });
    }
    void descramble(const float *Y_N1, float *Y_N2) { DVBS2HIP_CHK(ctx, dvbs2hip_pl_descramble(ctx->h, Y_N1, Y_N2, F())); }
};

// replaces Multiplier_AGC_cc_naive<R> (Multiplier_AGC_cc_naive.cpp:22-46; task "imultiply", sockets X_N / Z_N: Multiplier.hxx:49-60): N values per frame, as the reference's
// constructor takes them -- 2 * pl_frame_size for `mult_agc` (DVBS2.cpp:653-657), 2 * pl_frame_size * osf with energy 1 / osf for `front_agc` (DVBS2.cpp:660-664)
class Multiplier_AGC_hip : public Module_hip {
public:
    Multiplier_AGC_hip(std::shared_ptr<Context> c, int N, float output_energy = 1.f) : Module_hip(std::move(c), "Multiplier_AGC_hip"), N_(N), energy_(output_energy)
    {
        if (N <= 0 || N % 2) throw spu::tools::invalid_argument(__FILE__, __LINE__, __func__, "'N' has to be a positive even number of floats");
        auto &t = create_task("imultiply");
        auto s1 = create_socket_in<float>(t, "X_N", (size_t)N);
        auto s2 = create_socket_out<float>(t, "Z_N", (size_t)N);
        create_codelet(t, [s1, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Multiplier_AGC_hip &>(m).imultiply(tk[s1].template get_dataptr<const float>(), tk[s2].template get_dataptr<float>());
            return 0;
        });
    }
    void imultiply(const float *X_N, float *Z_N) { DVBS2HIP_CHK(ctx, dvbs2hip_agc_imultiply(ctx->h, X_N, Z_N, N_ / 2, energy_, F())); }
private:
    int N_;
    float energy_;
};

// replaces Synchronizer_freq_coarse<R> in the transmission phase: its task `synchronize` is the frequency shift alone (Synchronizer_freq_coarse_DVBS2_aib.cpp:43-50 ->
// Multiplier_sine_ccc_naive::imultiply; sockets Synchronizer_freq_coarse.hxx:35-39).  The PLL that finds the frequency (update_phase, one step per pilot symbol fed back from the
// timing synchronizer in the learning phases) is sample-serial and stays on the CPU: set_curr_freq() takes its result.  N values per frame (2 * pl_frame_size * osf: DVBS2.cpp build_synchronizer_freq_coarse)
template <typename R = float>
class Synchronizer_freq_coarse_hip : public Module_hip {
public:
    Synchronizer_freq_coarse_hip(std::shared_ptr<Context> c, int N) : Module_hip(std::move(c), "Synchronizer_freq_coarse_hip"), N_(N)
    {
        if (N <= 0 || N % 2) throw spu::tools::invalid_argument(__FILE__, __LINE__, __func__, "'N' has to be a positive even number of floats");
        auto &t = create_task("synchronize");
        auto sX = create_socket_in<R>(t, "X_N1", (size_t)N);
        auto sF = create_socket_out<R>(t, "FRQ", 1);
        auto sP = create_socket_out<R>(t, "PHS", 1);
        auto sY = create_socket_out<R>(t, "Y_N2", (size_t)N);
        create_codelet(t, [sX, sF, sP, sY](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Synchronizer_freq_coarse_hip &>(m).synchronize(tk[sX].template get_dataptr<const R>(), tk[sF].template get_dataptr<R>(), tk[sP].template get_dataptr<R>(),
                                                                       tk[sY].template get_dataptr<R>());
            return 0;
        });
    }
    void synchronize(const R *X_N1, R *FRQ, R *PHS, R *Y_N2) { DVBS2HIP_CHK(ctx, dvbs2hip_sync_coarse_synchronize(ctx->h, X_N1, FRQ, PHS, Y_N2, N_ / 2, F())); }
    void set_curr_freq(R estimated_freq) { DVBS2HIP_CHK(ctx, dvbs2hip_sync_coarse_set_freq(ctx->h, (float)estimated_freq)); }
    void reset() { DVBS2HIP_CHK(ctx, dvbs2hip_sync_coarse_reset(ctx->h)); }
private:
    int N_;
};

// replaces Framer<B>::remove_plh (Framer.hxx:330-343)
class Framer_hip : public Module_hip {
public:
    explicit Framer_hip(std::shared_ptr<Context> c) : Module_hip(std::move(c), "Framer_hip")
    {
        auto &t = create_task("remove_plh");
        auto s1 = create_socket_in<float>(t, "Y_N1", 2 * ctx->sz.pl_frame_sym);
        auto s2 = create_socket_out<float>(t, "Y_N2", 2 * ctx->sz.N_xfec_sym);
        create_codelet(t, [s1, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Framer_hip &>(m).remove_plh(tk[s1].template get_dataptr<const float>(), tk[s2].template get_dataptr<float>());
            return 0;
        });
    }
    void remove_plh(const float *Y_N1, float *Y_N2) { DVBS2HIP_CHK(ctx, dvbs2hip_remove_plh(ctx->h, Y_N1, Y_N2, F())); }
};

// replaces Scrambler_BB<D>::descramble (Scrambler_BB.hxx:51-72)
template <typename B = int>
class Scrambler_BB_hip : public Module_hip {
public:
    explicit Scrambler_BB_hip(std::shared_ptr<Context> c) : Module_hip(std::move(c), "Scrambler_BB_hip")
    {
        auto &t = create_task("descramble");
        auto s1 = create_socket_in<B>(t, "Y_N1", ctx->sz.K_bch);
        auto s2 = create_socket_out<B>(t, "Y_N2", ctx->sz.K_bch);
        create_codelet(t, [s1, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Scrambler_BB_hip &>(m).descramble(tk[s1].template get_dataptr<const B>(), tk[s2].template get_dataptr<B>());
            return 0;
        });
    }
    void descramble(const B *Y_N1, B *Y_N2) { DVBS2HIP_CHK(ctx, dvbs2hip_bb_descramble(ctx->h, (const int32_t *)Y_N1, (int32_t *)Y_N2, F())); }
};

// replaces Monitor_BFER<B> (built DVBS2.cpp:575-591): check_errors (bound TX_RX_BB/main.cpp:93-94) and check_errors2 (bound
// RX/main_sched.cpp:222-223, its BE / FE / BER / FER sockets feeding probes :244-247)
template <typename B = int>
class Monitor_BFER_hip : public Module_hip {
public:
    Monitor_BFER_hip(std::shared_ptr<Context> c, unsigned max_fe = 100) : Module_hip(std::move(c), "Monitor_BFER_hip"), max_fe(max_fe)
    {
        {
            auto &t = create_task("check_errors");
            auto sU = create_socket_in<B>(t, "U", ctx->sz.K_bch);
            auto sV = create_socket_in<B>(t, "V", ctx->sz.K_bch);
            create_codelet(t, [sU, sV](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
                static_cast<Monitor_BFER_hip &>(m).check_errors(tk[sU].template get_dataptr<const B>(), tk[sV].template get_dataptr<const B>());
                return 0;
            });
        }
        {
            auto &t = create_task("check_errors2");
            auto sU = create_socket_in<B>(t, "U", ctx->sz.K_bch);
            auto sV = create_socket_in<B>(t, "V", ctx->sz.K_bch);
            auto sFRA = create_socket_out<int64_t>(t, "FRA", 1);
            auto sBE = create_socket_out<int32_t>(t, "BE", 1);
            auto sFE = create_socket_out<int32_t>(t, "FE", 1);
            auto sBER = create_socket_out<float>(t, "BER", 1);
            auto sFER = create_socket_out<float>(t, "FER", 1);
            create_codelet(t, [sU, sV, sFRA, sBE, sFE, sBER, sFER](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
                static_cast<Monitor_BFER_hip &>(m).check_errors2(tk[sU].template get_dataptr<const B>(), tk[sV].template get_dataptr<const B>(),
                                                                 tk[sFRA].template get_dataptr<int64_t>(), tk[sBE].template get_dataptr<int32_t>(),
                                                                 tk[sFE].template get_dataptr<int32_t>(), tk[sBER].template get_dataptr<float>(),
                                                                 tk[sFER].template get_dataptr<float>());
                return 0;
            });
        }
    }
    spu::runtime::Task &operator[](mnt::tsk t) { return *tasks[(size_t)t]; }
    spu::runtime::Socket &operator[](mnt::sck::check_errors s) { return (*tasks[(size_t)mnt::tsk::check_errors])[(size_t)s]; }
    spu::runtime::Socket &operator[](mnt::sck::check_errors2 s) { return (*tasks[(size_t)mnt::tsk::check_errors2])[(size_t)s]; }
    void check_errors(const B *U, const B *V) { DVBS2HIP_CHK(ctx, dvbs2hip_monitor_check_errors(ctx->h, (const int32_t *)U, (const int32_t *)V, F())); }
    void check_errors2(const B *U, const B *V, int64_t *FRA, int32_t *BE, int32_t *FE, float *BER, float *FER)
    { DVBS2HIP_CHK(ctx, dvbs2hip_monitor_check_errors2(ctx->h, (const int32_t *)U, (const int32_t *)V, FRA, BE, FE, BER, FER, F())); }
    void get(uint64_t &fra, uint64_t &be, uint64_t &fe) { uint64_t c[3]; DVBS2HIP_CHK(ctx, dvbs2hip_monitor_get(ctx->h, c)); fra = c[0]; be = c[1]; fe = c[2]; }
    // tools::Monitor_reduction for one process per GPU (TX_RX_BB/main.cpp:123-125,155-161): RCCL sum over the ranks, collective
    void reduce_init(int rank, int world_size, const std::string &rendezvous_path, int timeout_ms = 60000)
    { DVBS2HIP_CHK(ctx, dvbs2hip_monitor_reduce_init(ctx->h, rank, world_size, rendezvous_path.c_str(), timeout_ms)); }
    void get_reduced(uint64_t &fra, uint64_t &be, uint64_t &fe) { uint64_t c[3]; DVBS2HIP_CHK(ctx, dvbs2hip_monitor_reduce(ctx->h, c)); fra = c[0]; be = c[1]; fe = c[2]; }
    bool is_done() { uint64_t a, b, c; get(a, b, c); return c >= max_fe; }      // stop at max_fe (DVBS2.cpp:136)
    bool is_done_all() { uint64_t a, b, c; get_reduced(a, b, c); return c >= max_fe; }   // Monitor_reduction::is_done_all
    void reset() { DVBS2HIP_CHK(ctx, dvbs2hip_monitor_reset(ctx->h)); }
private:
    unsigned max_fe;
};

// the fused RX chain (PL descramble ... BB descramble) as ONE task: intermediates stay on the GPU
template <typename B = int>
class Receiver_BB_hip : public Module_hip {
public:
    explicit Receiver_BB_hip(std::shared_ptr<Context> c) : Module_hip(std::move(c), "Receiver_BB_hip")
    {
        auto &t = create_task("receive");
        auto s1 = create_socket_in<float>(t, "Y_N1", 2 * ctx->sz.pl_frame_sym);
        auto sV = create_socket_out<B>(t, "V_K", ctx->sz.K_bch);
        auto sL = create_socket_out<int8_t>(t, "CWD_LDPC", 1);
        auto sB = create_socket_out<int8_t>(t, "CWD_BCH", 1);
        create_codelet(t, [s1, sV, sL, sB](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Receiver_BB_hip &>(m).receive(tk[s1].template get_dataptr<const float>(), tk[sV].template get_dataptr<B>(),
                                                      tk[sL].template get_dataptr<int8_t>(), tk[sB].template get_dataptr<int8_t>());
            return 0;
        });
    }
    void receive(const float *pl, B *V_K, int8_t *cwd_ldpc, int8_t *cwd_bch)
    {
        ctx->pin(pl, sizeof(float) * 2 * (size_t)ctx->sz.pl_frame_sym * F()); ctx->pin(V_K, sizeof(B) * (size_t)ctx->sz.K_bch * F());
        ctx->pin(cwd_ldpc, (size_t)F()); ctx->pin(cwd_bch, (size_t)F());
        DVBS2HIP_CHK(ctx, dvbs2hip_rx_bb(ctx->h, pl, nullptr, (int32_t *)V_K, cwd_ldpc, cwd_bch, F()));
    }
};

// replaces Synchronizer_frame_DVBS2_fast<R> (Synchronizer_frame_DVBS2_fast.cpp; tasks and sockets of
// Synchronizer_frame.hxx:40-98): three tasks on one module, state in the device context
template <typename R = float>
class Synchronizer_frame_hip : public Module_hip {
public:
    Synchronizer_frame_hip(std::shared_ptr<Context> c, R alpha = (R)0.9, R trigger = (R)30, int vec_width = 8)
    : Module_hip(std::move(c), "Synchronizer_frame_hip")
    {
        DVBS2HIP_CHK(ctx, dvbs2hip_sync_frame_set_params(ctx->h, (float)alpha, (float)trigger, vec_width));
        const size_t N = 2 * (size_t)ctx->sz.pl_frame_sym;
        {
            auto &t = create_task("synchronize");
            auto sX = create_socket_in<R>(t, "X_N1", N);
            auto sD = create_socket_out<int>(t, "DEL", 1);
            auto sF = create_socket_out<int>(t, "FLG", 1);
            auto sT = create_socket_out<R>(t, "TRI", 1);
            auto sY = create_socket_out<R>(t, "Y_N2", N);
            create_codelet(t, [sX, sD, sF, sT, sY](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
                static_cast<Synchronizer_frame_hip &>(m).synchronize(tk[sX].template get_dataptr<const R>(), tk[sD].template get_dataptr<int>(),
                                                                     tk[sF].template get_dataptr<int>(), tk[sT].template get_dataptr<R>(),
                                                                     tk[sY].template get_dataptr<R>());
                return 0;
            });
        }
        {
            auto &t = create_task("synchronize1");
            auto sX = create_socket_in<R>(t, "X_N1", N);
            auto s1 = create_socket_out<R>(t, "cor_SOF", N);
            auto s2 = create_socket_out<R>(t, "cor_PLSC", N);
            create_codelet(t, [sX, s1, s2](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
                static_cast<Synchronizer_frame_hip &>(m).synchronize1(tk[sX].template get_dataptr<const R>(), tk[s1].template get_dataptr<R>(),
                                                                      tk[s2].template get_dataptr<R>());
                return 0;
            });
        }
        {
            auto &t = create_task("synchronize2");
            auto sX = create_socket_in<R>(t, "X_N1", N);
            auto s1 = create_socket_in<R>(t, "cor_SOF", N);
            auto s2 = create_socket_in<R>(t, "cor_PLSC", N);
            auto sD = create_socket_out<int>(t, "DEL", 1);
            auto sF = create_socket_out<int>(t, "FLG", 1);
            auto sT = create_socket_out<R>(t, "TRI", 1);
            auto sY = create_socket_out<R>(t, "Y_N2", N);
            create_codelet(t, [sX, s1, s2, sD, sF, sT, sY](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
                static_cast<Synchronizer_frame_hip &>(m).synchronize2(tk[sX].template get_dataptr<const R>(), tk[s1].template get_dataptr<const R>(),
                                                                      tk[s2].template get_dataptr<const R>(), tk[sD].template get_dataptr<int>(),
                                                                      tk[sF].template get_dataptr<int>(), tk[sT].template get_dataptr<R>(),
                                                                      tk[sY].template get_dataptr<R>());
                return 0;
            });
        }
    }
    spu::runtime::Task &operator[](sfm::tsk t) { return *tasks[(size_t)t]; }
    spu::runtime::Socket &operator[](sfm::sck::synchronize s) { return (*tasks[(size_t)sfm::tsk::synchronize])[(size_t)s]; }
    spu::runtime::Socket &operator[](sfm::sck::synchronize1 s) { return (*tasks[(size_t)sfm::tsk::synchronize1])[(size_t)s]; }
    spu::runtime::Socket &operator[](sfm::sck::synchronize2 s) { return (*tasks[(size_t)sfm::tsk::synchronize2])[(size_t)s]; }
    void synchronize(const R *X_N1, int *DEL, int *FLG, R *TRI, R *Y_N2)
    { DVBS2HIP_CHK(ctx, dvbs2hip_sync_frame_synchronize(ctx->h, X_N1, (int32_t *)DEL, (int32_t *)FLG, TRI, Y_N2, F())); }
    void synchronize1(const R *X_N1, R *cor_SOF, R *cor_PLSC) { DVBS2HIP_CHK(ctx, dvbs2hip_sync_frame_synchronize1(ctx->h, X_N1, cor_SOF, cor_PLSC, F())); }
    void synchronize2(const R *X_N1, const R *cor_SOF, const R *cor_PLSC, int *DEL, int *FLG, R *TRI, R *Y_N2)
    { DVBS2HIP_CHK(ctx, dvbs2hip_sync_frame_synchronize2(ctx->h, X_N1, cor_SOF, cor_PLSC, (int32_t *)DEL, (int32_t *)FLG, TRI, Y_N2, F())); }
    void reset() { DVBS2HIP_CHK(ctx, dvbs2hip_sync_frame_reset(ctx->h)); }                        // Interface_reset
    R get_metric() { float m; int32_t f; DVBS2HIP_CHK(ctx, dvbs2hip_sync_frame_get_metric(ctx->h, &m, &f)); return (R)m; }
    bool get_packet_flag() { float m; int32_t f; DVBS2HIP_CHK(ctx, dvbs2hip_sync_frame_get_metric(ctx->h, &m, &f)); return f != 0; }
};

// replaces Synchronizer_Luise_Reggiannini_DVBS2_aib<R> (LR = true; lr_alpha as Factory/.../Synchronizer_freq_fine.hpp:25)
// and Synchronizer_freq_phase_DVBS2_aib<R> (LR = false): task and sockets of Synchronizer_freq_fine.hxx:32-47
template <typename R = float>
class Synchronizer_freq_fine_hip : public Module_hip {
public:
    Synchronizer_freq_fine_hip(std::shared_ptr<Context> c, bool luise_reggiannini, R lr_alpha = (R)0.999)
    : Module_hip(std::move(c), luise_reggiannini ? "Synchronizer_Luise_Reggiannini_hip" : "Synchronizer_freq_phase_hip"), lr(luise_reggiannini)
    {
        if (lr) DVBS2HIP_CHK(ctx, dvbs2hip_sync_lr_set_alpha(ctx->h, (float)lr_alpha));
        const size_t N = 2 * (size_t)ctx->sz.pl_frame_sym;
        auto &t = create_task("synchronize");
        auto sX = create_socket_in<R>(t, "X_N1", N);
        auto sF = create_socket_out<R>(t, "FRQ", 1);
        auto sP = create_socket_out<R>(t, "PHS", 1);
        auto sY = create_socket_out<R>(t, "Y_N2", N);
        create_codelet(t, [sX, sF, sP, sY](spu::module::Module &m, spu::runtime::Task &tk, size_t) -> int {
            static_cast<Synchronizer_freq_fine_hip &>(m).synchronize(tk[sX].template get_dataptr<const R>(), tk[sF].template get_dataptr<R>(),
                                                                     tk[sP].template get_dataptr<R>(), tk[sY].template get_dataptr<R>());
            return 0;
        });
    }
    void synchronize(const R *X_N1, R *FRQ, R *PHS, R *Y_N2)
    {
        if (lr) DVBS2HIP_CHK(ctx, dvbs2hip_sync_lr_synchronize(ctx->h, X_N1, FRQ, PHS, Y_N2, F()));
        else DVBS2HIP_CHK(ctx, dvbs2hip_sync_freq_phase_synchronize(ctx->h, X_N1, FRQ, PHS, Y_N2, F()));
    }
    void reset() { if (lr) DVBS2HIP_CHK(ctx, dvbs2hip_sync_lr_reset(ctx->h)); }
private:
    bool lr;
};

}  // namespace module
}  // namespace aff3ct
