// Development aid (CPU, no GPU needed): builds the LDPC plan of a MODCOD exactly as dvbs2hip_create does and prints what the
// static hybrid search chose -- LDS rows, slots per layer, whether the parity chain sits at the last two slots.
//   hipcc -std=c++17 -I dvbs2_amd/csrc -I include tools/plan_probe.cpp -L dvbs2_amd/lib -ldvbs2hip -Wl,-rpath,$PWD/dvbs2_amd/lib -o tools/bin/plan_probe
#include "dvbs2hip_internal.h"
#include "../include/dvbs2hip.h"
#include <cstdio>
using namespace dvbs2;
int main(int argc, char **argv)
{
    dvbs2hip_cfg cfg;
    if (dvbs2hip_cfg_from_modcod(argc > 1 ? argv[1] : "QPSK-N_8/9", &cfg)) { std::printf("unknown modcod\n"); return 1; }
    LdpcPlan pl;
    const std::string e = ldpc_build_plan(pl, cfg.N_ldpc, cfg.K_ldpc, cfg.ldpc_n_rows, cfg.ldpc_row_ptr, cfg.ldpc_addr, -1, 160 * 1024, argc > 2);
    std::printf("plan: '%s' fast %d deg %d mode %d wg8 %d dups_in_lds %d | LDS rows %d (info %d) global rows %d (info %d) | lds bytes %d gwork words %d\n", e.c_str(), pl.fast,
                pl.fast_deg, pl.fast_mode, pl.fast_wg8, pl.w8_dups_in_lds, pl.w8_nl, pl.w8_nl_info, pl.w8_ng, pl.w8_ng_info, pl.w8_lds_bytes, pl.w8_gwork_words);
    if (pl.fast_wg8 && (pl.fast_mode == 3 || pl.fast_mode == 4 || pl.fast_mode == 5)) {
        // the static hybrid's contract with the kernel: the first NL slots of every layer (and only those) are LDS accesses
        const int NL = pl.fast_mode == 3 ? 9 : ldpc_park_nl(pl.fast_mode);
        bool ok = true;
        for (int r = 0; r < pl.q; r++) for (int j = 0; j < pl.fast_deg; j++) ok &= (((pl.w8_tab[(size_t)r * LDPC_FAST_STRIDE + j] >> 29) & 1u) != 0u) == (j < NL);
        int swaps = 0;
        if (pl.fast_mode >= 4) for (size_t i = (size_t)pl.q * LDPC_FAST_STRIDE; i < pl.w8_tab.size(); i++) swaps += pl.w8_tab[i] != 0xFFu;
        std::printf("hybrid: %d LDS slots per layer %s | parked rows %d, row moves per iteration %d, swaps in the table %d\n", NL, ok ? "ok" : "BROKEN", pl.fast_mode >= 4 ? ldpc_park_nr(pl.fast_mode) : 0,
                    pl.w8_park_moves, swaps);
    }
    if (pl.fast_wg8) {
        for (int r = 0; r < pl.q; r++) {
            const uint32_t *T = &pl.w8_tab[(size_t)r * LDPC_FAST_STRIDE];
            const int ncf = (int)(T[28] & 0xFF);
            std::printf("layer %2d: ncf %d :", r, ncf);
            for (int i = 0; i < ncf; i++) std::printf(" (slot %u lvl %u shift %u)", T[48 + i] & 31u, T[48 + i] >> 8, (T[32 + i] & 0x7FFu) / 4);
            std::printf("  prim %07x\n", T[27]);
        }
    }
    return 0;
}
