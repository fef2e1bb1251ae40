// Development aid (CPU, no GPU needed): builds the LDPC plan of a MODCOD exactly as dvbs2hip_create does and prints what the
// static hybrid search chose -- LDS rows, slots per layer, whether the parity chain sits at the last two slots.
//   hipcc -std=c++17 -I dvbs2_amd/csrc -I include tools/plan_probe.cpp -L dvbs2_amd/lib -ldvbs2hip -Wl,-rpath,$PWD/dvbs2_amd/lib -o tools/bin/plan_probe
#include "dvbs2hip_internal.h"
#include "../include/dvbs2hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace dvbs2;
int main(int argc, char **argv)
{
    dvbs2hip_cfg cfg;
    if (dvbs2hip_cfg_from_modcod(argc > 1 ? argv[1] : "QPSK-N_8/9", &cfg)) { std::printf("unknown modcod\n"); return 1; }
    LdpcPlan pl;
    const std::string e = ldpc_build_plan(pl, cfg.N_ldpc, cfg.K_ldpc, cfg.ldpc_n_rows, cfg.ldpc_row_ptr, cfg.ldpc_addr, -1, argc > 3 ? atoi(argv[3]) : 160 * 1024, argc > 2 && argv[2][0] == 's');
    std::printf("plan: '%s' fast %d deg %d mode %d wg8 %d dups_in_lds %d | LDS rows %d (info %d) global rows %d (info %d) | lds bytes %d gwork words %d\n", e.c_str(), pl.fast,
                pl.fast_deg, pl.fast_mode, pl.fast_wg8, pl.w8_dups_in_lds, pl.w8_nl, pl.w8_nl_info, pl.w8_ng, pl.w8_ng_info, pl.w8_lds_bytes, pl.w8_gwork_words);
    if (pl.fast_wg8 && (pl.fast_mode == 3 || pl.fast_mode == 4 || pl.fast_mode == 5)) {
        // the static hybrid's contract with the kernel: the first NL slots of every layer (and only those) are LDS accesses
        const int NL = pl.fast_mode == 3 ? 9 : ldpc_park_nl(pl.fast_mode);
        bool ok = true;
        for (int r = 0; r < pl.q; r++) for (int j = 0; j < pl.fast_deg; j++) ok &= (((pl.w8_tab[(size_t)r * LDPC_FAST_STRIDE + j] >> 29) & 1u) != 0u) == (j < NL);
        int swaps = 0;
        if (pl.fast_mode >= 4) for (size_t i = (size_t)pl.q * LDPC_FAST_STRIDE; i < (size_t)pl.q * LDPC_FAST_STRIDE + (size_t)pl.q * ldpc_park_nr(pl.fast_mode); i++) swaps += pl.w8_tab[i] != 0xFFu;
        std::printf("hybrid: %d LDS slots per layer %s | parked rows %d, row moves per iteration %d, swaps in the table %d\n", NL, ok ? "ok" : "BROKEN", pl.fast_mode >= 4 ? ldpc_park_nr(pl.fast_mode) : 0,
                    pl.w8_park_moves, swaps);
    }
    if (!pl.nat_haz.empty()) {      // natural-order kernels: how many checks the hazard planes flag (plane 0: shares a bit with the check before it; plane 1: with one of the NAT_HAZ_WINDOW before it)
        const size_t hw = pl.nat_haz.size() / 2;
        int n0 = 0, n1 = 0;
        for (size_t i = 0; i < hw; i++) { n0 += __builtin_popcount(pl.nat_haz[i]); n1 += __builtin_popcount(pl.nat_haz[hw + i]); }
        std::printf("natural order: %d of %d checks share a bit with the check before them, %d with one of the %d before them\n", n0, pl.M, n1, NAT_HAZ_WINDOW);
    }
    if (pl.fast_cu1) {
        // mode 6 (k_ldpc_cu1.hip): replay one cycle of the tables as the kernel reads them -- where every slot's row is said to be (LDS position in THAT layer) against
        // a simulation of the row-keeping waves' swaps from the start state (w8_rows: rows at the positions, then the parity groups' positions, then the slots' rows)
        const int q = pl.q, NRT = 2 * ldpc_cu1_nrg(), P = pl.w8_nl;
        std::vector<int> lds(pl.w8_rows.begin(), pl.w8_rows.begin() + P), reg;
        for (int k = 0; k < NRT; k++) reg.push_back((int)pl.w8_rows[(size_t)P + pl.w8_ng + q + k]);
        const std::vector<int> lds0 = lds, reg0 = reg;
        const uint32_t *srv = &pl.w8_tab[(size_t)q * LDPC_FAST_STRIDE];
        bool ok = pl.w8_ng == 0 && (int)pl.w8_lds_junk == P * LDPC_Z * 4;
        int swaps = 0, dupmax = 0, seen_rows = 0;
        std::vector<char> seen(pl.n_groups, 0);
        for (int g : lds) if (g >= 0 && g < pl.n_groups && !seen[g]) { seen[g] = 1; seen_rows++; }
        for (int g : reg) if (g >= 0 && g < pl.n_groups && !seen[g]) { seen[g] = 1; seen_rows++; }
        for (int r = 0; r < q && ok; r++) {
            const uint32_t *T = &pl.w8_tab[(size_t)r * LDPC_FAST_STRIDE];
            // slots 25 / 26 are p_c / p_{c-1}; every slot is an LDS access whose base is the position of a row; a slot's row must not be under way
            std::vector<int> used;
            for (int j = 0; j < pl.fast_deg; j++) { const uint32_t base = (T[j] >> 11) & 0x3FFFFu; ok &= ((T[j] >> 29) & 1u) && base % (LDPC_Z * 4) == 0 && (int)(base / (LDPC_Z * 4)) < P; used.push_back(lds[base / (LDPC_Z * 4)]); }
            ok &= used[pl.fast_deg - 2] == pl.n_info + r && used[pl.fast_deg - 1] == pl.n_info + (r + q - 1) % q;
            const int ncf = (int)(T[28] & 0xFF);
            dupmax = std::max(dupmax, ncf);
            for (int i = 0; i < ncf; i++) ok &= (int)(T[48 + i] & 31u) == i && i < LDPC_CU1_HA && T[32 + i] == T[i] && !((T[27] >> i) & 1u);       // conflict entry i is slot i, in half A, not primary
            for (int j = ncf; j < pl.fast_deg; j++) ok &= ((T[27] >> j) & 1u) != 0u;                                                                // every other slot is primary
            // the next layer's rows must already be in place when this layer's swaps run (a swap may take the whole layer), and a swapped row is used by neither
            std::vector<int> next;
            { const uint32_t *Tn = &pl.w8_tab[(size_t)((r + 1) % q) * LDPC_FAST_STRIDE]; (void)Tn; }
            for (int k = 0; k < NRT; k++) {
                const uint32_t e = srv[(size_t)r * NRT + k];
                if (e == 0xFFu) continue;
                swaps++;
                ok &= (int)e < P;
                const int a = lds[e], b = reg[k];
                for (int g : used) ok &= g != a && g != b;
                lds[e] = b; reg[k] = a;
            }
        }
        ok &= lds == lds0 && reg == reg0 && seen_rows == pl.n_groups;
        for (int r = 0; r < q; r++) {      // where parity group r starts: its position, or 0xFFFFFFFF and then one of the register slots
            const uint32_t wh = pl.w8_rows[(size_t)P + r];
            if (wh == 0xFFFFFFFFu) ok &= std::find(reg0.begin(), reg0.end(), pl.n_info + r) != reg0.end();
            else ok &= wh % (LDPC_Z * 4) == 0 && lds0[wh / (LDPC_Z * 4)] == pl.n_info + r;
        }
        std::printf("cu1: positions %d pairs %d swaps per iteration %d max duplicate edges per layer %d tables %s\n", P, pl.cu1_pairs, swaps, dupmax, ok ? "ok" : "BROKEN");
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
