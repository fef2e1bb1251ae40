// a1 -- DVB-S2 LDPC decoder, horizontal-layered normalised min-sum, for gfx950.
//
// Replaces the decoder behind tools::Codec_LDPC<B,Q>::get_decoder_siho()
// (/root/reference src/common/Factory/DVBS2/DVBS2.cpp:418-449, type "BP_HORIZONTAL_LAYERED",
// implem "NMS").  Not a port of AFF3CT's SIMD decoder: that one sweeps the checks of one
// frame serially (inter-frame SIMD only).  Here ONE FRAME = ONE WORKGROUP of 6 wavefronts and
// the code's quasi-cyclic structure gives the intra-frame parallelism:
//
//   * the M = 360 q checks split into q LAYERS {c : c mod q = r}; inside a layer the 360
//     checks (t = c div q) touch every bit-group through a circulant: check t reads element
//     (t - t0) mod 360 of the group, so lane t's accesses are unit-stride across lanes
//     (conflict-free LDS banks / fully coalesced global segments);
//   * posteriors live on chip: as many 360-element bit-groups as fit are kept in LDS
//     (all of them for N = 16200); for N = 64800 (259 KB fp32 > 160 KB LDS) the
//     least-touched groups spill to a per-frame global workspace that stays L2/MALL hot;
//   * check->variable messages are stored COMPRESSED per check: the two output
//     magnitudes, 27 sign bits and the 5-bit position of the minimum = 12 bytes per check
//     instead of 4 bytes per edge, and decompress bit-exactly to the fp32 messages.
//
// This file holds (1) the host-side PLAN: layer tables, storage policy and the choice between the
// kernels, and (2) the GENERIC table-driven kernel (per-slot flags, hybrid LDS/global image), which is
// the fallback for codes the fast kernels reject -- check degree above 27 or more than 16 duplicate
// edges per layer -- and the subject of `DVBS2HIP_LDPC_PATH=generic` experiments.  Every DVB-S2 code
// shipped here runs on k_ldpc_wg8.hip (NMS / MS / SPA; mode 6 of the min-sum decoder: k_ldpc_cu1.hip).
//
// Schedule and arithmetic are restated in oracle/dvbs2_oracle.c (ORC_SCHED_QC) and the two
// must agree bit for bit: tests/test_ldpc_gpu.py.
//
// Tuning / experiment knobs (environment, read when a handle is created):
//   DVBS2HIP_LDPC_PATH=generic        force the generic kernel
//   DVBS2HIP_LDPC_FAST_MODE=lds|global|static|park|park4|cu1   posterior image of the fast kernels (default: lds for N = 16200; for N = 64800 the static hybrid with rows parked in the idle waves' registers (park), static = without them)
//   DVBS2HIP_LDPC_LOCK_DUPS=0         static hybrid without forcing the duplicate-edge bit-groups into LDS (then the generic kernel runs)
//   DVBS2HIP_LDPC_C2V=lds|global, DVBS2HIP_LDPC_LDS_GROUPS=n   generic kernel storage policy
//   DVBS2HIP_LDPC_BLOCKS_PER_CU, DVBS2HIP_LDPC_GRID_MAX, DVBS2HIP_LDS_LIMIT   occupancy / scaling experiments
#include "dvbs2hip_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace dvbs2 {

// ------------------------------------------------------------------------------------------
// host: layer tables
// ------------------------------------------------------------------------------------------
static const char *const PLAN_RETRY_GENERIC = "\x01generic";

// ------------------------------------------------------------------------------------------
// MODE 4 ("parked rows", k_ldpc_wg8.hip): on-chip bit-groups = LDS rows + rows PARKED in the registers of the workgroup's two
// idle waves while no layer needs them.  The schedule is static and cyclic over the q layers of an iteration:
//   * the on-chip set S gives every layer exactly NL (14 or 15) of its slots (k_ldpc_wg8.hip knows at compile time which slots
//     are LDS accesses);
//   * n_pos LDS positions, NR = |S| - n_pos of them each shared by a PAIR of rows (X, Y) with register slot k: while X is in the
//     position Y sits in the slot, and twice per iteration the idle waves swap them ("during layer s": between the end-of-layer
//     barriers s-1 and s).  A pair is compatible when two layers s1, s2 that use neither row separate the layers that use X
//     (all inside (s1, s2)) from those that use Y (all inside (s2, s1)): X is swapped in during s1 -- it is first needed by s1 + 1
//     at the earliest -- and out during s2, after its last use;
//   * the other rows of S own a position for good.  Pairs = a matching in the compatibility graph (randomised greedy, restarts).
// Everything here is host code; the result is verified by simulating one full cycle before it is used.
struct ParkPlan {
    std::vector<char> in_chip;              // [n_groups]
    std::vector<std::vector<int>> pos;      // [n_groups][q]: LDS position of the row during layer r (-1: not in LDS then)
    std::vector<uint32_t> srv;              // [q][NR]: LDS position slot k swaps with during layer r (0xFF: none)
    std::vector<int> lds0, reg0;            // state at the start of layer 0: bit-group at LDS position / in register slot (-1: empty)
    int n_pos = 0, nl0 = 0, n_moves = 0, n_pairs = 0;
};

static bool plan_parked(const std::vector<std::vector<int>> &mult, const std::vector<char> &banned, int q, int NL, int n_pos, int NRmax,
                        ParkPlan &out, std::string &why, const std::vector<char> *nopair = nullptr)
{
    const int n_groups = (int)mult.size();
    // the row-keeping waves hand their NRmax rows back through LDS positions 0 .. NRmax-1 at the end of a frame (w8_park_server, cu1_keeper): there have to be that many
    if (n_pos < NRmax) { why = "fewer LDS positions than parked rows (DVBS2HIP_LDS_LIMIT too small for this mode)"; return false; }
    uint32_t rng = 2463534242u;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng >> 4; };
    std::vector<int> touches(n_groups, 0);
    std::vector<char> dup(n_groups, 0);
    for (int g = 0; g < n_groups; g++) for (int r = 0; r < q; r++) { touches[g] += mult[g][r]; dup[g] |= mult[g][r] > 1; }
    for (int round = 0; round < 8; round++) {
        // ---- (A) the on-chip set: exactly NL slots of every layer, at most n_pos + NRmax rows, the doubly connected groups among them
        const int cap = n_pos + NRmax;
        std::vector<char> inS(n_groups, 0);
        std::vector<int> cnt(q, 0);
        int size = 0;
        auto fits = [&](int g) { for (int r = 0; r < q; r++) if (cnt[r] + mult[g][r] > NL) return false; return true; };
        auto add = [&](int g, int s) { inS[g] = s > 0; size += s; for (int r = 0; r < q; r++) cnt[r] += s * mult[g][r]; };
        for (int g = 0; g < n_groups; g++) if (dup[g]) {
            if (banned[g] || !fits(g) || size >= cap) { why = "doubly connected bit-groups do not fit"; return false; }
            add(g, +1);
        }
        for (;;) {
            int best = 0, bg = -1;
            for (int g = 0; g < n_groups && size < cap; g++) if (!inS[g] && !banned[g] && fits(g) && touches[g] * 16 + (int)(rnd() % 16) > best) { best = touches[g] * 16 + 15; bg = g; }
            if (bg < 0) break;
            add(bg, +1);
        }
        auto cost = [&]() { int c = 0; for (int r = 0; r < q; r++) c += (NL - cnt[r]) * (NL - cnt[r]); return c; };
        int c0 = cost();
        for (int it = 0; it < 600000 && c0 > 0; it++) {
            int g_out = -1, g_in = -1;
            if (rnd() & 1) { do { g_out = (int)(rnd() % n_groups); } while (!inS[g_out]); if (dup[g_out]) g_out = -1; }
            if (rnd() % 10 != 0) { do { g_in = (int)(rnd() % n_groups); } while (inS[g_in] || banned[g_in]); }
            if (g_out >= 0) add(g_out, -1);
            bool ok = true;
            if (g_in >= 0) { ok = fits(g_in) && size < cap; if (ok) add(g_in, +1); }
            const int c1 = ok ? cost() : 1 << 30;
            if (ok && (c1 <= c0 || rnd() % 500 == 0)) c0 = c1;
            else { if (ok && g_in >= 0) add(g_in, -1); if (g_out >= 0) add(g_out, +1); }
        }
        if (c0 != 0) { why = "no on-chip set with the same number of slots in every layer"; continue; }
        std::vector<int> S;
        for (int g = 0; g < n_groups; g++) if (inS[g]) S.push_back(g);
        const int need = std::max(0, (int)S.size() - n_pos);    // pairs (= register slots) needed
        if (need > NRmax) { why = "on-chip set too large"; continue; }
        // ---- (B) compatible pairs and their swap layers
        struct Edge { int x, y, s1, s2; };                      // x in LDS during (s1, s2), y during (s2, s1)
        std::vector<Edge> edges;
        auto used_by = [&](int g, int l) { return mult[g][((l % q) + q) % q] > 0; };
        for (size_t i = 0; i < S.size(); i++) for (size_t j = i + 1; j < S.size(); j++) {
            const int x = S[i], y = S[j];
            if (nopair && ((*nopair)[x] || (*nopair)[y])) continue;          // rows that own their position for good (mode 6: the parity groups)
            int bs1 = -1, bs2 = -1, bsc = -1;
            for (int s1 = 0; s1 < q; s1++) {
                if (used_by(x, s1) || used_by(y, s1) || used_by(y, s1 + 1)) continue;          // y leaves during s1: not needed in s1 nor s1 + 1 .. ; x arrives
                for (int d = 1; d < q; d++) {
                    const int s2 = (s1 + d) % q;
                    if (used_by(x, s2) || used_by(y, s2) || used_by(x, s2 + 1)) continue;
                    bool okp = true;
                    for (int l = 0; l < q && okp; l++) {
                        const bool inside = ((l - s1) % q + q) % q < d;                      // l in [s1, s2)
                        if (used_by(x, l) && !inside) okp = false;
                        if (used_by(y, l) && inside) okp = false;
                    }
                    if (!okp) continue;
                    // slack: layers between the swap and the first use after it (the more, the less a late swap can delay a layer)
                    int f1 = 1, f2 = 1;
                    while (!used_by(x, s1 + f1)) f1++;
                    while (!used_by(y, s2 + f2)) f2++;
                    const int sc = std::min(f1, f2);
                    if (sc > bsc) { bsc = sc; bs1 = s1; bs2 = s2; }
                }
            }
            if (bs1 >= 0) edges.push_back({x, y, bs1, bs2});
        }
        // ---- (C) matching: randomised greedy, low-degree rows first; if that falls short, a MAXIMUM matching (Edmonds' blossom algorithm, started from the greedy
        //      one: the graph has a few hundred vertices) -- mode 6 needs 72 disjoint pairs among the 160 information rows of the N = 64800 8/9 code and greedy finds 71
        std::vector<int> best_match;
        for (int attempt = 0; attempt < 3000 && (int)best_match.size() < need; attempt++) {
            std::vector<int> deg(n_groups, 0), order(edges.size());
            std::vector<uint32_t> key(edges.size());
            for (const Edge &e : edges) { deg[e.x]++; deg[e.y]++; }
            for (size_t i = 0; i < edges.size(); i++) { order[i] = (int)i; key[i] = (uint32_t)(deg[edges[i].x] + deg[edges[i].y]) * 64u + rnd() % (attempt == 0 ? 1u : 512u); }
            std::sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
            std::vector<char> taken(n_groups, 0);
            std::vector<int> m;
            for (int i : order) if (!taken[edges[i].x] && !taken[edges[i].y]) { taken[edges[i].x] = taken[edges[i].y] = 1; m.push_back(i); }
            if (m.size() > best_match.size()) best_match = m;
            if (attempt >= 40 && need > 45) break;      // (many pairs wanted: leave the rest to the exact algorithm)
        }
        if ((int)best_match.size() < need) {
            const int V = n_groups;
            std::vector<std::vector<int>> adj(V);
            std::vector<std::vector<int>> eid(V, std::vector<int>(V, -1));
            for (size_t i = 0; i < edges.size(); i++) { adj[edges[i].x].push_back(edges[i].y); adj[edges[i].y].push_back(edges[i].x); eid[edges[i].x][edges[i].y] = eid[edges[i].y][edges[i].x] = (int)i; }
            std::vector<int> mate(V, -1), par(V), base(V), qu;
            std::vector<char> used(V), blossom(V);
            for (int i : best_match) { mate[edges[i].x] = edges[i].y; mate[edges[i].y] = edges[i].x; }
            auto lca = [&](int a, int b) {
                std::vector<char> seen(V, 0);
                for (;;) { a = base[a]; seen[a] = 1; if (mate[a] < 0) break; a = par[mate[a]]; }
                for (;;) { b = base[b]; if (seen[b]) return b; b = par[mate[b]]; }
            };
            auto mark_path = [&](int v, int b, int child) {
                while (base[v] != b) { blossom[base[v]] = blossom[base[mate[v]]] = 1; par[v] = child; child = mate[v]; v = par[mate[v]]; }
            };
            auto find_path = [&](int root) -> int {
                std::fill(used.begin(), used.end(), 0); std::fill(par.begin(), par.end(), -1);
                for (int i = 0; i < V; i++) base[i] = i;
                qu.clear(); qu.push_back(root); used[root] = 1;
                for (size_t qh = 0; qh < qu.size(); qh++) {
                    const int v = qu[qh];
                    for (int to : adj[v]) {
                        if (base[v] == base[to] || mate[v] == to) continue;
                        if (to == root || (mate[to] >= 0 && par[mate[to]] >= 0)) {
                            const int cb = lca(v, to);
                            std::fill(blossom.begin(), blossom.end(), 0);
                            mark_path(v, cb, to); mark_path(to, cb, v);
                            for (int i = 0; i < V; i++) if (blossom[base[i]]) { base[i] = cb; if (!used[i]) { used[i] = 1; qu.push_back(i); } }
                        } else if (par[to] < 0) {
                            par[to] = v;
                            if (mate[to] < 0) return to;
                            used[mate[to]] = 1; qu.push_back(mate[to]);
                        }
                    }
                }
                return -1;
            };
            for (int v = 0; v < V; v++) if (mate[v] < 0 && !adj[v].empty()) {
                int u = find_path(v);
                while (u >= 0) { const int pv = par[u], ppv = mate[pv]; mate[u] = pv; mate[pv] = u; u = ppv; }
            }
            best_match.clear();
            for (int v = 0; v < V; v++) if (mate[v] > v) best_match.push_back(eid[v][mate[v]]);
        }
        if (getenv("DVBS2HIP_VERBOSE")) fprintf(stderr, "[dvbs2hip] plan_parked: %zu rows on chip, %zu compatible pairs, matching %zu of %d needed\n", S.size(), edges.size(), best_match.size(), need);
        if ((int)best_match.size() < need) { why = "not enough compatible pairs of rows"; continue; }
        best_match.resize((size_t)need);
        // ---- (D) tables: positions 0 .. need-1 are the shared ones (slot k <-> position k), then the rows that own theirs
        out.in_chip = inS; out.n_pos = n_pos; out.n_pairs = need; out.n_moves = 4 * need;
        out.pos.assign(n_groups, std::vector<int>(q, -1));
        out.srv.assign((size_t)q * NRmax, 0xFFu);
        out.lds0.assign(n_pos, -1); out.reg0.assign(NRmax, -1);
        std::vector<char> paired(n_groups, 0);
        for (int k = 0; k < need; k++) {
            const Edge &e = edges[best_match[k]];
            paired[e.x] = paired[e.y] = 1;
            const int d = ((e.s2 - e.s1) % q + q) % q;
            for (int l = 0; l < q; l++) {
                const int off = ((l - e.s1) % q + q) % q;
                if (off >= 1 && off < d) out.pos[e.x][l] = k;            // x: layers strictly between s1 and s2
                if (off > d) out.pos[e.y][l] = k;                       // y: strictly between s2 and s1
            }
            out.srv[(size_t)e.s1 * NRmax + k] = (uint32_t)k;
            out.srv[(size_t)e.s2 * NRmax + k] = (uint32_t)k;
            // start of layer 0 (before its swaps): x holds the position iff 0 is in (s1, s2]
            const int o0 = ((0 - e.s1) % q + q) % q;
            const bool x_in = o0 >= 1 && o0 <= d;
            out.lds0[k] = x_in ? e.x : e.y; out.reg0[k] = x_in ? e.y : e.x;
        }
        int P = need;
        for (int g : S) if (!paired[g]) { for (int l = 0; l < q; l++) out.pos[g][l] = P; out.lds0[P] = g; P++; }
        if (P > n_pos) { why = "internal: positions"; return false; }
        out.nl0 = P;
        // ---- (E) one full cycle simulated from that state: every access finds its row, no swap touches a row in use, the state closes
        {
            std::vector<int> lds = out.lds0, reg = out.reg0;
            for (int r = 0; r < q; r++) {
                for (int g : S) if (mult[g][r]) { const int Pg = out.pos[g][r]; if (Pg < 0 || lds[Pg] != g) { why = "simulation: row not where the table says"; return false; } }
                for (int k = 0; k < NRmax; k++) {
                    const uint32_t e = out.srv[(size_t)r * NRmax + k];
                    if (e == 0xFFu) continue;
                    const int a = lds[e], b = reg[k];
                    if (a < 0 || b < 0) { why = "simulation: swap with an empty place"; return false; }
                    if (mult[a][r] || mult[a][(r + 1) % q] || mult[b][r]) { why = "simulation: swap of a row in use"; return false; }
                    lds[e] = b; reg[k] = a;
                }
            }
            if (lds != out.lds0 || reg != out.reg0) { why = "simulation: the cycle does not close"; return false; }
        }
        return true;
    }
    return false;
}

static std::string build_plan_impl(LdpcPlan &pl, int N, int K, int n_rows, const int32_t *row_ptr,
                                   const int32_t *addr, int lds_groups_req, size_t lds_limit, int spa_rule, bool allow_fast, bool small_batch)
{
    const bool spa = spa_rule != 0;
    pl.spa = spa; pl.spa_rule = spa_rule;
    if (N <= 0 || K <= 0 || K >= N) return "LDPC: need 0 < K < N";
    const int M = N - K;
    if (M % LDPC_Z || K % LDPC_Z) return "LDPC: N-K and K must be multiples of 360";
    if (n_rows != K / LDPC_Z) return "LDPC: address table must have K/360 rows";
    const int q = M / LDPC_Z;
    pl.N = N; pl.K = K; pl.M = M; pl.q = q; pl.n_info = n_rows; pl.n_groups = n_rows + q;
    for (int i = 0; i < row_ptr[n_rows]; i++)
        if (addr[i] < 0 || addr[i] >= M) return "LDPC: address out of range";

    struct Slot { int group, t0, lvl, mask0; };
    std::vector<std::vector<Slot>> layers(q);
    for (int g = 0; g < n_rows; g++)
        for (int p = row_ptr[g]; p < row_ptr[g + 1]; p++) {
            const int r = addr[p] % q, t0 = addr[p] / q;
            int lvl = 0;
            for (const Slot &s : layers[r]) {
                if (s.group == g) { lvl++; if (s.t0 == t0) return "LDPC: duplicate edge in address table"; }
            }
            if (lvl > 3) return "LDPC: more than 4 edges of one bit-group in one layer";
            layers[r].push_back({g, t0, lvl, 0});
        }
    for (int r = 0; r < q; r++) {
        layers[r].push_back({n_rows + r, 0, 0, 0});                 // p_c
        if (r > 0) layers[r].push_back({n_rows + r - 1, 0, 0, 0});  // p_{c-1}, same t
        else       layers[r].push_back({n_rows + q - 1, 1, 0, 1});  // p_{c-1} = group q-1, element t-1; absent for c = 0
    }
    pl.deg_max = 0; pl.E = -1;
    pl.layer_deg.assign(q, 0); pl.layer_lvl.assign(q, 0);
    for (int r = 0; r < q; r++) {
        pl.layer_deg[r] = (int)layers[r].size();
        pl.deg_max = std::max(pl.deg_max, pl.layer_deg[r]);
        pl.E += LDPC_Z * pl.layer_deg[r];
        for (const Slot &s : layers[r]) pl.layer_lvl[r] = std::max(pl.layer_lvl[r], s.lvl);
    }
    if (pl.deg_max > LDPC_MAX_SLOTS) return "LDPC: check degree > 27 not supported by the packed message format";

    // ---- storage policy: which bit-groups live in LDS
    std::vector<int> touches(pl.n_groups, 0);
    for (int r = 0; r < q; r++) for (const Slot &s : layers[r]) touches[s.group]++;
    std::vector<int> order(pl.n_groups);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return touches[a] > touches[b]; });

    const size_t c2v_bytes = (size_t)M * 12, grp_bytes = (size_t)LDPC_Z * 4;
    const char *env_c2v = getenv("DVBS2HIP_LDPC_C2V");
    const char *env_grp = getenv("DVBS2HIP_LDPC_LDS_GROUPS");
    if (env_grp) lds_groups_req = atoi(env_grp);
    const size_t all_post = (size_t)pl.n_groups * grp_bytes;
    bool c2v_lds;
    if (env_c2v) c2v_lds = !strcmp(env_c2v, "lds");
    else c2v_lds = (all_post + c2v_bytes <= lds_limit);     // everything on chip when it fits
    if (c2v_lds && c2v_bytes + grp_bytes > lds_limit) c2v_lds = false;
    const size_t avail = lds_limit - (c2v_lds ? c2v_bytes : 0);
    int max_groups = (int)std::min<size_t>(pl.n_groups, avail / grp_bytes);
    int n_lds = (lds_groups_req < 0) ? max_groups : std::min(lds_groups_req, max_groups);
    pl.lds_groups = n_lds; pl.c2v_lds = c2v_lds; pl.hybrid = n_lds < pl.n_groups;

    pl.groups.assign(pl.n_groups, {0, 0});
    int nl = 0, ng = 0;
    for (int i = 0; i < pl.n_groups; i++) {
        const int g = order[i];
        if (i < n_lds) pl.groups[g] = {(uint32_t)(nl++ * LDPC_Z), 1u};
        else           pl.groups[g] = {(uint32_t)(ng++ * LDPC_Z), 0u};
    }
    pl.lds_post_words = nl * LDPC_Z;
    pl.glb_post_words = ng * LDPC_Z;
    pl.gwork_words = pl.glb_post_words + (c2v_lds ? 0 : 3 * M);
    pl.lds_bytes = (size_t)pl.lds_post_words * 4 + (c2v_lds ? c2v_bytes : 0);

    // ---- regular-code fast path: uniform check degree (11 or 27), few same-layer duplicates
    {
        // slots per layer in the unrolled kernel: 11 or 27 when every check has that degree, else the
        // layers are padded to 13 / 27 slots with NULL slots that read a row of +inf and store nowhere
        bool regular = pl.deg_max <= 27, uniform = true;
        int maxc = 0;
        for (int r = 0; r < q && regular; r++) {
            if (pl.layer_deg[r] != pl.deg_max) uniform = false;
            int c = 0;
            for (const Slot &s : layers[r]) c += s.lvl > 0;
            maxc = std::max(maxc, c);
        }
        const char *env_path = getenv("DVBS2HIP_LDPC_PATH");
        if ((env_path && !strcmp(env_path, "generic")) || !allow_fast) regular = false;
        if (regular && maxc <= LDPC_FAST_MAXC) {
            pl.fast = true;
            pl.fast_deg = (uniform && pl.deg_max == 11) ? 11 : (uniform && pl.deg_max == 27) ? 27 : pl.deg_max <= 13 ? 13 : 27;
            pl.fast_pad = !(uniform && pl.deg_max == pl.fast_deg);
            const int xrows = pl.fast_pad ? 1 : 0;        // the +inf row
            // posterior image: LDS when a CU holds two frames of it (N = 16200), else the workgroup's global slot,
            // upgraded below to the static hybrid where the code allows.  DVBS2HIP_LDPC_FAST_MODE=lds|global|static forces one.
            const char *env_mode = getenv("DVBS2HIP_LDPC_FAST_MODE");
            pl.fast_mode = ((size_t)(pl.n_groups + 1 + xrows) * grp_bytes * 2 <= lds_limit + 1024) ? 0 : 1;
            const bool env_hyb = env_mode && (!strcmp(env_mode, "static") || !strcmp(env_mode, "park") || !strcmp(env_mode, "park4"));      // static: hybrid without parked rows; park: the default for the long codes
            if (env_mode && !env_hyb) pl.fast_mode = (!strcmp(env_mode, "lds") && pl.fast_mode == 0) ? 0 : 1;
            std::vector<uint32_t> gbase(pl.n_groups), glds(pl.n_groups, 0u);
            int n_l = 0, n_g = 0;
            // mode 3 (STATIC hybrid, normal frames): pick the LDS-resident bit-groups so that EVERY layer has
            // exactly NL = 9 of its 27 slots in LDS; the kernel then knows at compile time which slots are LDS
            // accesses.  Greedy fill + randomised local search on sum_r (NL - count_r)^2 (deterministic seed).
            std::vector<char> in_lds(pl.n_groups, 0);
            ParkPlan park;
            // mode 6 (k_ldpc_cu1.hip): ONE frame per CU, the whole posterior image on chip -- n_pos LDS positions + the rows parked in the registers of the
            // workgroup's four row-keeping waves (two groups of two waves, ldpc_cu1_nrg() rows each); every slot of every layer is an LDS access.  Parity groups
            // pair like information groups (the 160 information rows of the N = 64800 8/9 code have a maximum matching of 71 pairs, 72 are needed): a parity row
            // that starts an iteration in a register slot is loaded by its row-keeping wave (a stride-q gather), the others by the working waves' scatter.
            const bool env_cu1 = env_mode && !strcmp(env_mode, "cu1");
            // (round 6) ... and of the min-sum decoder on a handle made for at most one frame per CU (`small_batch`): a call is then one frame's ten iterations on one CU, and
            // with two lanes per check those take 0.42 ms instead of 0.54 (F = 1 .. 256, same box; bit-identical results)
            if ((env_cu1 || (!env_mode && !spa && (LDPC_CU1_DEFAULT || small_batch)) || (!env_mode && spa && LDPC_CU1_SPA_DEFAULT)) && spa_rule != 2 && pl.fast_mode == 1 && pl.fast_deg == 27 && !pl.fast_pad) {
                const int n_pos = ((int)lds_limit - LDPC_CU1_XCHG_BYTES - 128) / (int)grp_bytes - 1;      // [positions | junk row | exchange area | misc]
                std::vector<std::vector<int>> mult(pl.n_groups, std::vector<int>(q, 0));
                for (int r = 0; r < q; r++) for (const Slot &sl : layers[r]) mult[sl.group][r]++;
                std::vector<char> banned(pl.n_groups, 0);
                std::string why = "more rows than positions and register slots";
                int maxdup = 0;
                for (int r = 0; r < q; r++) { int c = 0; for (const Slot &sl : layers[r]) c += sl.lvl > 0; maxdup = std::max(maxdup, c); }
                const int need = pl.n_groups - n_pos;
                if (n_pos >= 2 * q && need <= 2 * ldpc_cu1_nrg() && maxdup <= LDPC_CU1_HA - 2 && need > 0 &&
                    plan_parked(mult, banned, q, pl.fast_deg, n_pos, 2 * ldpc_cu1_nrg(), park, why) && park.n_pairs == need) {
                    pl.fast_mode = 6;
                    for (int g = 0; g < pl.n_groups; g++) in_lds[g] = 1;
                    pl.w8_dups_in_lds = true;
                } else if (getenv("DVBS2HIP_VERBOSE") || env_cu1) fprintf(stderr, "[dvbs2hip] LDPC plan: mode 6 (one frame per CU) not used (%s)\n", why.c_str());
            }
            {
                const int NL = 9;
                const bool want = pl.fast_mode != 6 && (env_hyb || env_cu1 || (!env_mode && pl.fast_mode == 1));
                if (want && pl.fast_deg == 27 && !pl.fast_pad) {
                    const int cap = (int)(lds_limit / 2 / grp_bytes) - 1;            // two frames per CU, one junk row each
                    std::vector<std::vector<int>> mult(pl.n_groups, std::vector<int>(q, 0));
                    for (int r = 0; r < q; r++) for (const Slot &sl : layers[r]) mult[sl.group][r]++;
                    // the parity groups stay out of LDS: table order then puts p_c and p_{c-1} at the last two (global) slots of every layer,
                    // where k_ldpc_wg8.hip forwards the parity chain in a register; p_{c-1} of layer 0 is the absent-for-check-0 slot
                    auto banned_g = [&](int g) { return g >= n_rows; };
                    std::vector<int> cnt(q, 0);
                    int size = 0;
                    auto fits = [&](int g) { for (int r = 0; r < q; r++) if (cnt[r] + mult[g][r] > NL) return false; return true; };
                    auto add = [&](int g, int s) { in_lds[g] = s > 0; size += s; for (int r = 0; r < q; r++) cnt[r] += s * mult[g][r]; };
                    // bit-groups with two edges in one layer go in first and stay: the duplicate-edge replay and the
                    // store redirection of k_ldpc_wg8.hip then never leave LDS (DVBS2HIP_LDPC_LOCK_DUPS=0 to compare)
                    std::vector<char> locked(pl.n_groups, 0);
                    {
                        const char *el = getenv("DVBS2HIP_LDPC_LOCK_DUPS");
                        bool lock_ok = !(el && atoi(el) == 0);
                        std::vector<int> dups;
                        for (int g = 0; g < pl.n_groups && lock_ok; g++) {
                            bool d = false;
                            for (int r = 0; r < q; r++) d |= mult[g][r] > 1;
                            if (d) { if (banned_g(g)) lock_ok = false; dups.push_back(g); }
                        }
                        for (int g : dups) { if (!lock_ok) break; if (fits(g) && size < cap) { add(g, +1); locked[g] = 1; } else lock_ok = false; }
                        if (!lock_ok) { for (int g = 0; g < pl.n_groups; g++) if (in_lds[g]) add(g, -1); std::fill(locked.begin(), locked.end(), 0); }
                        pl.w8_dups_in_lds = lock_ok;
                    }
                    for (;;) {
                        int best = 0, bg = -1;
                        for (int i = 0; i < pl.n_groups && size < cap; i++) {
                            const int g = order[i];
                            if (in_lds[g] || banned_g(g) || !fits(g)) continue;
                            if (touches[g] > best) { best = touches[g]; bg = g; }
                        }
                        if (bg < 0) break;
                        add(bg, +1);
                    }
                    auto cost = [&]() { int c = 0; for (int r = 0; r < q; r++) c += (NL - cnt[r]) * (NL - cnt[r]); return c; };
                    uint32_t rng = 12345u;
                    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
                    int c0 = cost();
                    for (int it = 0; it < 400000 && c0 > 0; it++) {
                        int g_out = -1, g_in = -1;
                        if (rnd() & 1) { do { g_out = (int)(rnd() % pl.n_groups); } while (!in_lds[g_out]); if (locked[g_out]) g_out = -1; }
                        if (rnd() % 10 != 0) { do { g_in = (int)(rnd() % pl.n_groups); } while (in_lds[g_in] || banned_g(g_in)); }
                        if (g_out >= 0) add(g_out, -1);
                        bool ok = true;
                        if (g_in >= 0) { ok = fits(g_in) && size < cap; if (ok) add(g_in, +1); }
                        const int c1 = ok ? cost() : 1 << 30;
                        if (ok && (c1 <= c0 || rnd() % 500 == 0)) c0 = c1;
                        else { if (ok && g_in >= 0) add(g_in, -1); if (g_out >= 0) add(g_out, +1); }
                    }
                    if (c0 == 0) pl.fast_mode = 3;
                    else { std::fill(in_lds.begin(), in_lds.end(), 0); pl.w8_dups_in_lds = false; }
                    // modes 4 / 5: rows parked in the idle waves' registers on top of the LDS rows (DVBS2HIP_LDPC_FAST_MODE=static keeps mode 3, park4 mode 4
                    // for the min-sum kernel too)
                    if (pl.fast_mode == 3 && pl.w8_dups_in_lds && !(env_mode && !strcmp(env_mode, "static"))) {
                        std::vector<char> banned(pl.n_groups, 0);
                        for (int g = 0; g < pl.n_groups; g++) banned[g] = banned_g(g);
                        std::string why;
                        for (int m = (spa || (env_mode && !strcmp(env_mode, "park4"))) ? 4 : 5; m >= 4 && pl.fast_mode == 3; m--) {
                            if (plan_parked(mult, banned, q, ldpc_park_nl(m), cap, ldpc_park_nr(m), park, why)) {
                                pl.fast_mode = m;
                                for (int g = 0; g < pl.n_groups; g++) in_lds[g] = park.in_chip[g];
                            } else if (getenv("DVBS2HIP_VERBOSE")) fprintf(stderr, "[dvbs2hip] LDPC plan: mode %d (parked rows) not used (%s)\n", m, why.c_str());
                        }
                    }
                }
            }
            const bool cu1 = pl.fast_mode == 6;
            const bool parked = pl.fast_mode == 4 || pl.fast_mode == 5 || cu1, hyb = pl.fast_mode == 3 || parked;       // static hybrid: the first NLH slots of every layer are LDS accesses
            const int NLH = cu1 ? pl.fast_deg : parked ? ldpc_park_nl(pl.fast_mode) : 9, NRH = cu1 ? 2 * ldpc_cu1_nrg() : ldpc_park_nr(pl.fast_mode);
            if (hyb) {
                for (int g = 0; g < pl.n_groups; g++) {
                    if (in_lds[g]) { gbase[g] = (uint32_t)(n_l++ * LDPC_Z); glds[g] = 1u; }       // (mode 4: the LDS position of a row depends on the layer, park.pos)
                    else gbase[g] = (uint32_t)(n_g++ * LDPC_Z);
                }
            } else
                for (int g = 0; g < pl.n_groups; g++) { gbase[g] = (uint32_t)(g * LDPC_Z); glds[g] = pl.fast_mode == 0 ? 1u : 0u; }
            for (int g = 0; g < pl.n_groups; g++) pl.groups[g] = {gbase[g], glds[g]};
            // +inf row: LDS image = [groups | junk row | inf row]; global image = [groups | inf row]
            const uint32_t inf_row_words = (uint32_t)((pl.n_groups + (pl.fast_mode == 0 ? 1 : 0)) * LDPC_Z);
            // k_ldpc_wg8.hip image layout -- LDS: [rows | junk | +inf]; global: [junk | +inf | rows]
            pl.w8_tab.assign((size_t)q * LDPC_FAST_STRIDE, 0u);
            const int w8_lrows = pl.fast_mode == 0 ? pl.n_groups : pl.fast_mode == 3 ? n_l : parked ? park.n_pos : 0;
            bool park_bad = false, kd_ok = true;
            auto pack8 = [&](const Slot &sl, int r) -> uint32_t {
                if (sl.group < 0) return (uint32_t)((pl.fast_mode == 0 ? (w8_lrows + 1) * LDPC_Z * 4 : LDPC_Z * 4)) << 11;      // the +inf row
                const bool il = pl.fast_mode == 0 || (hyb && glds[sl.group]);
                uint32_t base = il ? gbase[sl.group] * 4u : 2u * LDPC_Z * 4u + gbase[sl.group] * 4u;
                if (il && parked) { const int P = park.pos[sl.group][r]; if (P < 0) park_bad = true; base = (uint32_t)(P < 0 ? 0 : P) * LDPC_Z * 4u; }
                return (uint32_t)(sl.t0 * 4) | (base << 11) | (il ? (1u << 29) : 0u);
            };
            for (int r = 0; r < q; r++) {
                uint32_t *T8 = &pl.w8_tab[(size_t)r * LDPC_FAST_STRIDE];
                uint32_t prim = 0, dupmask = 0; int nc = 0;
                // slot order: EARLY slots first (bit-group not touched by the previous layer, cyclically),
                // then the late ones; the absent-for-check-0 parity slot stays last.  Conflict levels were
                // fixed above in table order and travel with the slot.
                std::vector<char> prev_touch(pl.n_groups, 0);
                for (const Slot &sl : layers[(r + q - 1) % q]) prev_touch[sl.group] = 1;      // (layers[] hold real slots only)
                std::vector<Slot> ord;
                if (cu1) {      // one frame per CU: duplicate edges first (level, then table order; conflict entry i is slot i: all in the first half-check's slots), then the
                                // other information slots, p_c and p_{c-1} last (parity chain forwarded in a register by the second half-check's lanes)
                    for (int lvl = 1; lvl <= 3; lvl++) for (const Slot &sl : layers[r]) if (sl.lvl == lvl) ord.push_back(sl);
                    for (const Slot &sl : layers[r]) if (sl.lvl == 0 && sl.group < n_rows) ord.push_back(sl);
                    for (const Slot &sl : layers[r]) if (sl.lvl == 0 && sl.group >= n_rows) ord.push_back(sl);
                    const size_t nn = ord.size();
                    if ((int)nn != pl.fast_deg || ord[nn - 2].group != n_rows + r || ord[nn - 1].group != n_rows + (r + q - 1) % q || ord[nn - 2].t0 != 0 || (r > 0 && ord[nn - 1].t0 != 0))
                        return "LDPC: internal: mode 6 needs p_c and p_{c-1} at the last two slots";
                } else if (hyb) {      // static hybrid: the LDS-resident slots first (exactly 9 of them; 14 with parked rows), then the others
                    // (sum-product kernel: the duplicate edges first, in the order of the conflict list, so that conflict entry i is slot i as in the LDS-only image)
                    if (spa) for (int lvl = 1; lvl <= 3; lvl++) for (const Slot &sl : layers[r]) if (in_lds[sl.group] && sl.lvl == lvl) ord.push_back(sl);
                    for (const Slot &sl : layers[r]) if (in_lds[sl.group] && !(spa && sl.lvl > 0)) ord.push_back(sl);
                    if ((int)ord.size() != NLH) return "LDPC: internal: static hybrid balance broken";
                    for (const Slot &sl : layers[r]) if (!in_lds[sl.group]) ord.push_back(sl);
                    const size_t nn = ord.size();
                    if (nn < 2 || ord[nn - 2].group != n_rows + r || ord[nn - 1].group != n_rows + (r + q - 1) % q || ord[nn - 2].t0 != 0 || (r > 0 && ord[nn - 1].t0 != 0))
                        return "LDPC: internal: static hybrid needs p_c and p_{c-1} at the last two slots (parity chain forwarding)";
                } else {
                    // duplicate edges first (the only slots whose stores are redirected: ldpc_w8_kd), never the masked slot (a parity group: no duplicates)
                    // in the order of the conflict list (level, then table order): conflict entry i is slot i, which the sum-product layer relies on
                    for (int lvl = 1; lvl <= 3; lvl++) for (const Slot &sl : layers[r]) if (sl.lvl == lvl) ord.push_back(sl);
                    for (const Slot &sl : layers[r]) if (sl.lvl == 0 && !prev_touch[sl.group]) ord.push_back(sl);
                    for (const Slot &sl : layers[r]) if (sl.lvl == 0 && prev_touch[sl.group]) ord.push_back(sl);
                }
                if (!ord.empty() && layers[r].back().mask0 && !ord.back().mask0) return "LDPC: internal: masked slot must stay last";
                // NULL slots (group -1): behind the duplicate edges at the front of the layer in the LDS-only image (like those, their stores are redirected:
                // ldpc_w8_kd), else in front of the last real slot; the last real slot keeps position fast_deg-1 either way
                {
                    size_t nd = 0;
                    while (nd < ord.size() && ord[nd].lvl > 0) nd++;
                    while ((int)ord.size() < pl.fast_deg) ord.insert(pl.fast_mode == 0 ? ord.begin() + (long)nd : ord.end() - 1, Slot{-1, 0, 0, 0});
                }
                // conflict list sorted by level
                for (int lvl = 1; lvl <= 3; lvl++)
                    for (size_t j = 0; j < ord.size(); j++)
                        if (ord[j].lvl == lvl && ord[j].group >= 0) {
                            T8[32 + nc] = pack8(ord[j], r);
                            T8[48 + nc] = (uint32_t)j | ((uint32_t)lvl << 8);
                            dupmask |= 1u << j;
                            if (hyb && !glds[ord[j].group]) pl.w8_dups_in_lds = false;
                            nc++;
                        }
                for (size_t j = 0; j < ord.size(); j++) {
                    // byte shift (11 bits) | byte offset of the bit-group in its store (18 bits) | LDS flag
                    T8[j] = pack8(ord[j], r);
                    if (ord[j].lvl == 0 && ord[j].group >= 0) prim |= 1u << j;
                }
                // ncf | slot of entry 0 << 8 | level << 13 | slot of entry 1 << 16 | level << 21 ; entries 0 and 1 ; slots with a duplicate edge
                T8[27] = prim; T8[28] = (uint32_t)nc; T8[31] = dupmask;
                // contract with k_ldpc_wg8.hip: from slot ldpc_w8_kd(deg) on every slot is a primary edge (no redirected store, no NULL slot)
                // (the LDS-only image; with the hybrid image the same trick measured 0.7 % SLOWER on the 15 LDS slots of a normal-frame layer and is not used)
                for (int j = ldpc_w8_kd(pl.fast_deg); j < pl.fast_deg && pl.fast_mode == 0; j++) if (!((prim >> j) & 1u)) kd_ok = false;
                for (int i = 0; i < 2 && i < nc; i++) {
                    T8[28] |= ((T8[48 + i] & 31u) | ((T8[48 + i] >> 8) << 5)) << (8 + 8 * i);
                    T8[29 + i] = T8[32 + i];
                }
                if (nc > 0 && (T8[48] >> 8) != 1u) return "LDPC: internal: first conflict entry is not of level 1";
                if (pl.fast_mode == 0 || spa || cu1) for (int i = 0; i < nc; i++) if ((T8[48 + i] & 31u) != (uint32_t)i) return "LDPC: internal: conflict entry i is not slot i";
                if (spa) {
                    // the oracle's edge order of a check (information bits in address-table order, p_c, p_{c-1} = layers[r] as built above) as slots: the tanh-product
                    // rule multiplies in THAT order (fp32 products do not commute bit for bit); NULL slots (tanh(inf / 2) = 1, exact) fill the tail
                    std::vector<int> perm;
                    std::vector<char> used(ord.size(), 0);
                    for (const Slot &sl : layers[r])
                        for (size_t j = 0; j < ord.size(); j++)
                            if (!used[j] && ord[j].group == sl.group && ord[j].t0 == sl.t0) { perm.push_back((int)j); used[j] = 1; break; }
                    if ((int)perm.size() != pl.layer_deg[r] || nc > LDPC_TANH_ORDER - 48) return "LDPC: internal: edge-order table";
                    for (size_t j = 0; j < ord.size(); j++) if (!used[j]) perm.push_back((int)j);
                    for (size_t c = 0; c < perm.size(); c++) T8[LDPC_TANH_ORDER + c / 6] |= (uint32_t)perm[c] << (5 * (c % 6));
                }
            }
            if (!spa && !cu1 && (pl.fast_mode == 0 || LDPC_ATAB_HYB)) {      // per-lane address table of the min-sum layer (k_ldpc_wg8.hip, W8_ATAB: every slot of the LDS-only image; -DW8_ATAB_HYB: the LDS slots of the hybrid images)
                const int NW4 = (pl.fast_deg + 3) / 4;
                pl.w8_atab.assign((size_t)q * NW4 * LDPC_AT_LANES * 4, 0u);
                for (int r = 0; r < q; r++) {
                    const uint32_t *T8 = &pl.w8_tab[(size_t)r * LDPC_FAST_STRIDE];
                    for (int j = 0; j < 4 * NW4; j++)
                        for (int t = 0; t < LDPC_AT_LANES; t++) {
                            uint32_t v = 0x7FFFF000u;                                        // lanes past the 360th check / padding slots: an offset every buffer access drops
                            if (j < pl.fast_deg) {
                                const uint32_t e = T8[j], shift = e & 0x7FFu, base = (e >> 11) & 0x3FFFFu;
                                const bool il = pl.fast_mode == 0 || (hyb && j < NLH);            // (what the kernel takes for an LDS slot: w8_slot_lds; a NULL slot's entry carries no flag)
                                if (t < LDPC_Z) { const uint32_t d = ((uint32_t)t * 4u + (uint32_t)LDPC_Z * 4u - shift) % ((uint32_t)LDPC_Z * 4u); v = il ? d + base : d; }
                                else if (il) v = base;                                      // an LDS slot of an idle lane: any address inside the allocation (never accessed: `act`)
                            }
                            pl.w8_atab[(((size_t)r * NW4 + j / 4) * LDPC_AT_LANES + t) * 4 + (j & 3)] = v;
                        }
                }
            }
            if (spa && !cu1 && pl.fast_mode == 0 && LDPC_SPA_AT16 && pl.fast_deg <= LDPC_SPA_AT16_MAXDEG) {
                // (round 5) per-lane address table of the SUM-PRODUCT layer on the LDS-only image (k_ldpc_wg8.hip, W8_SPA_AT16): two 16-bit LDS byte addresses per dword (slot 2 k in the low
                // half), [q][pieces of 16 bytes][LDPC_AT_LANES][4] -- the image's 45 rows end at byte 64800, so every address of a real slot fits; a NULL slot reads the +inf WORD the
                // kernel keeps at junk row + 4 (the +inf row itself lies beyond 64 KB; the junk row is written at its word 0 only in this form), and the slot of check 0's absent
                // p_{c-1} points at the junk row's word 0 (its value is replaced by +inf, its store lands there)
                const int ND = (pl.fast_deg + 1) / 2, NP = (ND + 3) / 4;
                const uint32_t junk = (uint32_t)(w8_lrows * LDPC_Z * 4), infw = junk + 4u, inf_row = (uint32_t)((w8_lrows + 1) * LDPC_Z * 4);
                bool fits = junk + 8u <= 65536u;
                pl.w8_atab.assign((size_t)q * NP * LDPC_AT_LANES * 4, 0u);
                for (int r = 0; r < q && fits; r++) {
                    const uint32_t *T8 = &pl.w8_tab[(size_t)r * LDPC_FAST_STRIDE];
                    for (int j = 0; j < pl.fast_deg; j++)
                        for (int t = 0; t < LDPC_AT_LANES; t++) {
                            const uint32_t e = T8[j], shift = e & 0x7FFu, base = (e >> 11) & 0x3FFFFu;
                            uint32_t v = junk;                                                   // idle lanes: never accessed (`act`)
                            if (t < LDPC_Z) {
                                if (base == inf_row) v = infw;
                                else if (j == pl.fast_deg - 1 && r == 0 && t == 0) v = junk;
                                else v = ((uint32_t)t * 4u + (uint32_t)LDPC_Z * 4u - shift) % ((uint32_t)LDPC_Z * 4u) + base;
                            }
                            if (v >= 65536u) fits = false;
                            pl.w8_atab[(((size_t)r * NP + (j / 2) / 4) * LDPC_AT_LANES + t) * 4 + ((j / 2) & 3)] |= v << (16 * (j & 1));
                        }
                }
                if (!fits) return PLAN_RETRY_GENERIC;      // (no DVB-S2 code: an LDS-only image is 45 rows)
            }
            {   // image rows in storage order: LDS rows then global rows (bit-groups ascend inside each: info first)
                std::vector<int> lrow, grow;
                for (int g = 0; g < pl.n_groups; g++) ((pl.fast_mode == 0 || (hyb && glds[g])) ? lrow : grow).push_back(g);
                if (parked) {      // LDS rows = the positions that hold a row at the start of an iteration (layer 0), in position order
                    if (park_bad) return "LDPC: internal: parked-row table";
                    lrow.assign(park.lds0.begin(), park.lds0.begin() + park.nl0);
                }
                pl.w8_nl = (int)lrow.size(); pl.w8_ng = (int)grow.size();
                pl.w8_nl_info = cu1 ? (int)lrow.size() : (int)std::count_if(lrow.begin(), lrow.end(), [&](int g) { return g < pl.n_info; });      // (mode 6: parity rows may sit among the pairs' positions; the kernel looks at every position)
                pl.w8_ng_info = (int)std::count_if(grow.begin(), grow.end(), [&](int g) { return g < pl.n_info; });
                pl.w8_rows.clear();
                for (int g : lrow) pl.w8_rows.push_back((uint32_t)g);
                for (int g : grow) pl.w8_rows.push_back((uint32_t)g);
                // then, for the frame input of the parity part: where parity group r (bit-group n_info + r) lives --
                // byte offset of its row in LDS, or bit 31 | byte offset inside the workgroup's global slot ([junk][+inf][rows])
                for (int r = 0; r < q; r++) {
                    const int g = pl.n_info + r;
                    const bool in_lds = pl.fast_mode == 0 || (hyb && glds[g]);
                    if (in_lds && parked && !cu1) return "LDPC: internal: parity group among the parked rows";
                    if (cu1) {      // where the parity group is at the start of an iteration: byte offset of its LDS position, or 0xFFFFFFFF = in a register slot
                        uint32_t where = 0xFFFFFFFFu;
                        for (int P = 0; P < park.nl0; P++) if (park.lds0[P] == g) where = (uint32_t)(P * LDPC_Z * 4);
                        pl.w8_rows.push_back(where);
                        continue;
                    }
                    pl.w8_rows.push_back(in_lds ? (uint32_t)gbase[g] * 4u : 0x80000000u | (uint32_t)((2 * LDPC_Z + (int)gbase[g]) * 4));
                }
                if (parked) {      // then the bit-group in register slot k of the idle waves at the start of an iteration (0xFFFFFFFF: empty)
                    for (int k = 0; k < NRH; k++) pl.w8_rows.push_back(park.reg0[k] < 0 ? 0xFFFFFFFFu : (uint32_t)park.reg0[k]);
                    // and the idle waves' swaps behind the layer tables: [q][NR] x LDS position (0xFF: none)
                    pl.w8_tab.insert(pl.w8_tab.end(), park.srv.begin(), park.srv.end());
                    if (!cu1) {     // modes 4 / 5 (round 4): the same swaps as ONE 64-bit mask per layer ([q][lo, hi]; bit k = slot k swaps with LDS position k during layer r) -- the
                                    // row-keeping waves test a bit per slot instead of loading and comparing a table entry per slot (39 dependent scalar loads per layer)
                        for (int r = 0; r < q; r++) {
                            unsigned long long m = 0;
                            for (int k = 0; k < NRH; k++) {
                                const uint32_t e = park.srv[(size_t)r * NRH + k];
                                if (e == 0xFFu) continue;
                                if ((int)e != k) return "LDPC: internal: parked rows: pair k is expected at position k";
                                m |= 1ull << k;
                            }
                            pl.w8_tab.push_back((uint32_t)m); pl.w8_tab.push_back((uint32_t)(m >> 32));
                        }
                    }
                    if (cu1) {      // k_ldpc_cu1.hip reads the swaps as bit masks: [q][2 groups][lo, hi], bit k = slot k of the group swaps with its position (= its index) during layer r
                        const int NRG = ldpc_cu1_nrg();
                        for (int r = 0; r < q; r++) for (int gk = 0; gk < 2; gk++) {
                            unsigned long long m = 0;
                            for (int k = 0; k < NRG; k++) {
                                const uint32_t e = park.srv[(size_t)r * NRH + gk * NRG + k];
                                if (e == 0xFFu) continue;
                                if ((int)e != gk * NRG + k) return "LDPC: internal: mode 6 expects pair k at position k";
                                m |= 1ull << k;
                            }
                            pl.w8_tab.push_back((uint32_t)m); pl.w8_tab.push_back((uint32_t)(m >> 32));
                        }
                    }
                } else
                    for (size_t i = 0; i < lrow.size(); i++) if ((int)gbase[lrow[i]] != (int)i * LDPC_Z) return "LDPC: internal: LDS row order";
                for (size_t i = 0; i < grow.size(); i++) if ((int)gbase[grow[i]] != (int)i * LDPC_Z) return "LDPC: internal: global row order";
                const int n_lds_rows = parked ? park.n_pos : pl.w8_nl;
                pl.w8_lds_junk = (uint32_t)(n_lds_rows * LDPC_Z * 4);
                pl.w8_lds_bytes = (n_lds_rows + 1 + (pl.fast_pad && pl.fast_mode == 0 ? 1 : 0)) * LDPC_Z * 4 + LDPC_W8_MISC_BYTES;
                pl.w8_park_moves = parked ? park.n_moves : 0;
                pl.w8_st_base = (uint32_t)((2 + pl.w8_ng) * LDPC_Z * 4);
                pl.w8_gwork_words = (2 + pl.w8_ng) * LDPC_Z + 3 * M;
                if (cu1) {      // LDS: [positions | junk row | exchange area of the two half-checks | misc]; global: the packed state alone, 16 bytes per check {c1, c2, pk of half A, pk of half B}
                    pl.w8_lds_bytes = (n_lds_rows + 1) * LDPC_Z * 4 + LDPC_CU1_XCHG_BYTES + 128;
                    pl.w8_st_base = 0u;
                    pl.w8_gwork_words = 4 * M;
                    pl.cu1_pairs = park.n_pairs;
                }
                if (spa) pl.w8_gwork_words = (2 + pl.w8_ng) * LDPC_Z + pl.fast_deg * M;      // SPA: one fp32 message per edge slot, [layer][slot][360]
                if (spa && cu1) pl.w8_gwork_words = pl.fast_deg * M;                          // mode 6: the messages alone, [layer][half][group of 4 slots][360][4]
                {   // DVBS2HIP_LDPC_SLOT_ALIGN / _PAD (bytes): where a workgroup's slot starts -- measured without effect (docs/negative_results.md), kept for experiments
                    const char *ea = getenv("DVBS2HIP_LDPC_SLOT_ALIGN"), *ep = getenv("DVBS2HIP_LDPC_SLOT_PAD");
                    const size_t al = ea ? (size_t)atoi(ea) / 4 : 1, pad = ep ? (size_t)atoi(ep) / 4 : 0;
                    if (al > 1) pl.w8_gwork_words = (int)(((size_t)pl.w8_gwork_words + al - 1) / al * al);
                    pl.w8_gwork_words += (int)pad;
                }
            }
            // workspace of one workgroup: [posteriors kept in global memory | packed c->v state 3 M words]
            pl.glb_post_words = pl.fast_mode == 1 ? (pl.n_groups + xrows) * LDPC_Z : hyb ? n_g * LDPC_Z : 0;
            pl.lds_post_words = pl.fast_mode == 0 ? (pl.n_groups + 1 + xrows) * LDPC_Z : pl.fast_mode == 3 ? (n_l + 1) * LDPC_Z : parked ? (park.n_pos + 1) * LDPC_Z : 0;
            if (cu1) pl.glb_post_words = 0;
            pl.fast_inf_row = pl.fast_pad ? (int)(inf_row_words * 4u) : -1;
            pl.gwork_words = pl.glb_post_words + (spa ? pl.fast_deg * M : cu1 ? 4 * M : 3 * M);      // SPA: one fp32 message per edge slot
            pl.lds_bytes = (size_t)pl.lds_post_words * 4;
            pl.hybrid = hyb; pl.c2v_lds = false; pl.lds_groups = pl.fast_mode == 0 ? pl.n_groups : n_l;
            {   // k_ldpc_nat.hip (natural row order, one lane per frame): per layer the info slots (NULL-padded), then p_c, then p_{c-1}
                pl.nat_tab.assign((size_t)q * pl.fast_deg * 2, 0u);
                for (int r = 0; r < q; r++) {
                    uint32_t *T = &pl.nat_tab[(size_t)r * pl.fast_deg * 2];
                    const std::vector<Slot> &ls = layers[r];           // table order: info edges, p_c, p_{c-1}
                    const int n_real = (int)ls.size(), n_null = pl.fast_deg - n_real;
                    int j = 0;
                    auto put = [&](const Slot &sl) {
                        const bool par = sl.group >= pl.n_info;
                        T[2 * j] = (uint32_t)sl.t0 | (par ? 1u << 16 : 0u);
                        T[2 * j + 1] = par ? (uint32_t)(K + (sl.group - pl.n_info)) : (uint32_t)(sl.group * LDPC_Z);
                        j++;
                    };
                    for (int i = 0; i < n_real - 2; i++) put(ls[i]);
                    for (int i = 0; i < n_null; i++) { T[2 * j] = 1u << 17; T[2 * j + 1] = 0u; j++; }
                    put(ls[n_real - 2]); put(ls[n_real - 1]);
                    if (ls[n_real - 2].group != pl.n_info + r || !(ls[n_real - 1].group >= pl.n_info)) return "LDPC: internal: parity slots are not last";
                }
                // consecutive checks (cyclically) that share a bit other than the forwarded p_{c-1}
                pl.nat_haz.assign((size_t)(M + 31) / 32, 0u);
                auto vars_of = [&](int c, std::vector<int> &out) {
                    out.clear();
                    const int r = c % q, t = c / q;
                    for (const Slot &sl : layers[r]) {
                        if (sl.mask0 && c == 0) continue;
                        const int e = ((t - sl.t0) % LDPC_Z + LDPC_Z) % LDPC_Z;
                        out.push_back(sl.group < pl.n_info ? sl.group * LDPC_Z + e : K + q * e + (sl.group - pl.n_info));
                    }
                };
                // second plane (behind the first): check c shares a bit with one of the NAT_HAZ_WINDOW checks before it (cyclically) -- the kernels that request a
                // check's posteriors several checks ahead (k_ldpc_nat.hip, ldpc_nat_part_kernel: NAT_AHEAD checks) must not do so for these: the checks in between
                // have not written yet, and the stores of the one or two before them may still be in flight
                const size_t hw = pl.nat_haz.size();
                pl.nat_haz.resize(2 * hw, 0u);
                std::vector<int> a, b;
                for (int c = 0; c < M; c++) {
                    vars_of(c, a);
                    const int fwd_bit = c > 0 ? K + c - 1 : -1;
                    for (int d = 1; d <= NAT_HAZ_WINDOW; d++) {
                        vars_of(((c - d) % M + M) % M, b);
                        bool hz = false;
                        for (int x : a) if (x != fwd_bit && std::find(b.begin(), b.end(), x) != b.end()) hz = true;
                        if (hz && d == 1) pl.nat_haz[c >> 5] |= 1u << (c & 31);
                        if (hz) pl.nat_haz[hw + (c >> 5)] |= 1u << (c & 31);
                    }
                }
            }
            // one frame per 8-wave workgroup, two independent workgroups per CU (k_ldpc_wg8.hip); a code it cannot take (a static hybrid
            // whose doubly connected bit-groups do not all fit in LDS) goes to the generic table-driven kernel below
            {
                const bool w8_ok = (pl.fast_mode == 0 || pl.fast_mode == 1 || (hyb && pl.w8_dups_in_lds)) && kd_ok;
                if (!(w8_ok && (size_t)pl.w8_lds_bytes <= lds_limit + 512) || (spa && maxc > LDPC_SPA_MAXC)) return PLAN_RETRY_GENERIC;
                pl.fast_wg8 = true; pl.gwork_words = pl.w8_gwork_words; pl.fast_cu1 = cu1;
            }
        }
    }
    if (pl.n_groups > 255) return "LDPC: more than 255 bit-groups not supported by the packed entry format";
    pl.ent_stride = pl.deg_max <= 13 ? 13 : LDPC_MAX_SLOTS;
    // padding entries read slot 0 of a store that exists and are ignored
    const LdpcEntry null_entry = LE_NULL | (nl > 0 ? LE_LDS : 0u);
    pl.entries.assign((size_t)q * pl.ent_stride, null_entry);
    for (int r = 0; r < q; r++)
        for (size_t j = 0; j < layers[r].size(); j++) {
            const Slot &s = layers[r][j];
            const LdpcGroup &gl = pl.groups[s.group];
            pl.entries[(size_t)r * pl.ent_stride + j] =
                (LdpcEntry)s.t0 | ((gl.base / LDPC_Z) << LE_SLOT_SHIFT) | (gl.lds ? LE_LDS : 0u) |
                (s.mask0 ? LE_MASK0 : 0u) | ((uint32_t)s.lvl << LE_LVL_SHIFT);
        }
    if (spa && !pl.fast) return "LDPC: SPA is only implemented for codes the fast path accepts (check degree <= 27, at most 6 duplicate edges per layer)";
    return "";
}

std::string ldpc_build_plan(LdpcPlan &pl, int N, int K, int n_rows, const int32_t *row_ptr,
                            const int32_t *addr, int lds_groups_req, size_t lds_limit, int spa_rule, bool small_batch)
{
    std::string e = build_plan_impl(pl, N, K, n_rows, row_ptr, addr, lds_groups_req, lds_limit, spa_rule, true, small_batch);
    if (e == PLAN_RETRY_GENERIC) { pl = LdpcPlan(); e = build_plan_impl(pl, N, K, n_rows, row_ptr, addr, lds_groups_req, lds_limit, spa_rule, false, small_batch); }
    return e;
}

// ------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------
// LDS pointers carry their address space in the type, so the optimiser can never merge an
// LDS access and a global access into one flat access through a selected generic pointer.
typedef __attribute__((address_space(3))) float lds_float;
// The layer tables are read-only for the whole launch.  Reading them through the CONSTANT
// address space lets the compiler use scalar loads (SGPRs, scalar cache) for these
// wave-uniform addresses; through a plain global pointer it must assume the kernel's own
// stores may alias them and falls back to per-lane vector loads with a full memory round
// trip in front of every edge.
typedef const __attribute__((address_space(4))) uint32_t *const_u32;
typedef const __attribute__((address_space(4))) int32_t *const_i32;
typedef const __attribute__((address_space(4))) unsigned long long *const_u64;
static_assert(sizeof(LdpcGroup) == 8, "group table is read as 64-bit scalars");

__device__ __forceinline__ LdpcGroup group_ld(const_u64 groups, int g)
{
    const unsigned long long raw = groups[g];
    LdpcGroup v;
    v.base = (uint32_t)raw; v.lds = (uint32_t)(raw >> 32);
    return v;
}

// element (t - t0) mod 360 of the entry's bit-group, as a word offset into its store
__device__ __forceinline__ int ent_off(LdpcEntry e, int t)
{
    const int m = t - (int)(e & LE_T0_MASK);
    return (int)(((e >> LE_SLOT_SHIFT) & LE_SLOT_MASK) * LDPC_Z) + (int)min((unsigned)m, (unsigned)(m + LDPC_Z));
}
template <bool HYBRID>
__device__ __forceinline__ float post_ld(LdpcEntry e, int off, const lds_float *lpost, const float *gpost)
{
    if (HYBRID && !(e & LE_LDS)) return gpost[off];
    return lpost[off];
}
template <bool HYBRID>
__device__ __forceinline__ void post_st(LdpcEntry e, int off, lds_float *lpost, float *gpost, float v)
{
    if (HYBRID && !(e & LE_LDS)) gpost[off] = v;
    else lpost[off] = v;
}
// the edge does not exist for this lane (padding entry, or p_{c-1} of check 0)
__device__ __forceinline__ bool ent_absent(LdpcEntry e, int t) { return (e & LE_NULL) || ((e & LE_MASK0) && t == 0); }
__device__ __forceinline__ int ent_lvl(LdpcEntry e) { return (int)((e >> LE_LVL_SHIFT) & LE_LVL_MASK); }

// fp32 message from the packed per-check state: magnitude c1 at the slot of the minimum, c2
// elsewhere, sign bit j of pk
__device__ __forceinline__ float c2v_unpack(float c1, float c2, uint32_t pk, int j)
{
    const float mag = ((pk >> 27) == (uint32_t)j) ? c1 : c2;
    return __uint_as_float(__float_as_uint(mag) | ((pk << (31 - j)) & 0x80000000u));
}

template <int DEG, bool HYBRID, bool C2V_LDS>
__global__ void __launch_bounds__(LDPC_THREADS, (DEG > 13 && HYBRID) ? 2 : 3)      // (the 27-slot hybrid form needs ~172 registers: two waves per SIMD instead of three, no spills)
ldpc_layered_nms_kernel(const LdpcKParams p)
{
    extern __shared__ float smem[];
    lds_float *lpost = (lds_float *)smem;
    const int t = threadIdx.x;
    const bool act = t < LDPC_Z;
    const int M = p.M, q = p.q;
    const const_u32 entries = (const_u32)p.entries;
    const const_i32 layer_lvl = (const_i32)p.layer_lvl;
    const const_u64 groups = (const_u64)p.groups;

    for (int f = blockIdx.x; f < p.n_frames; f += gridDim.x) {
        const float *Y = p.llr + (size_t)f * p.N;
        float *gwork = p.gwork + (size_t)blockIdx.x * p.gwork_words;   // per-WORKGROUP slot: stays cache-hot across frames
        float *gpost = gwork;
        // packed c->v state [r][t]: two magnitudes + (5-bit min position | 27 sign bits).
        // Kept as two separately typed pointers (never a generic LDS-or-global pointer).
        lds_float *lc = lpost + p.lds_post_words;     // LDS image   (C2V_LDS)
        float *gc = gwork + p.glb_post_words;         // global image (!C2V_LDS)
#define C2V_LD(arr, i) (C2V_LDS ? lc[(arr) * M + (i)] : gc[(arr) * M + (i)])
#define C2V_ST(arr, i, val) do { if (C2V_LDS) lc[(arr) * M + (i)] = (val); else gc[(arr) * M + (i)] = (val); } while (0)

        // ---- load channel LLRs into the posterior stores (parity bits regrouped [r][t])
        if (act)
            for (int g = 0; g < p.n_groups; g++) {
                const LdpcGroup gl = group_ld(groups, g);
                const int src = g < p.n_info ? g * LDPC_Z + t : p.K + q * t + (g - p.n_info);
                const float v = Y[src];
                if (HYBRID && !gl.lds) gpost[gl.base + t] = v; else lpost[gl.base + t] = v;
            }
        for (int i = t; i < 3 * M; i += LDPC_THREADS) C2V_ST(0, i, 0.f);
        __syncthreads();

        int it = 0;
        bool ok = false;
        // the packed state of check (r, t) is private to lane t: prefetch the next layer's
        // while the current layer computes (global-memory latency off the critical path)
        float nx1 = 0.f, nx2 = 0.f, nxk = 0.f;
        if (act) { nx1 = C2V_LD(0, t); nx2 = C2V_LD(1, t); nxk = C2V_LD(2, t); }
        while (it < p.n_ite) {
            for (int r = 0; r < q; r++) {
                // the whole layer's table in SGPRs up front (unconditional, padded table)
                LdpcEntry E[DEG];
#pragma unroll
                for (int j = 0; j < DEG; j++) E[j] = entries[r * p.ent_stride + j];
                const int maxlvl = layer_lvl[r];
                const int ci = r * LDPC_Z + t;
                float v[DEG];
                float cst1 = 0.f, cst2 = 0.f, mn1 = INFINITY, mn2 = INFINITY;
                const float c1o = nx1, c2o = nx2;
                const uint32_t pko = __float_as_uint(nxk);
                uint32_t sacc = 0u;
                if (act) {
                    // ---- pass 1a: issue every posterior load of the check before using any
#pragma unroll
                    for (int j = 0; j < DEG; j++) v[j] = post_ld<HYBRID>(E[j], ent_off(E[j], t), lpost, gpost);
                    {
                        const int cn = (r + 1 < q ? ci + LDPC_Z : t);
                        nx1 = C2V_LD(0, cn); nx2 = C2V_LD(1, cn); nxk = C2V_LD(2, cn);
                    }
                    // ---- pass 1b: v->c = posterior - old c->v ; running min1/min2/sign
#pragma unroll
                    for (int j = 0; j < DEG; j++) {
                        float x = v[j] - c2v_unpack(c1o, c2o, pko, j);
                        if (ent_absent(E[j], t)) x = INFINITY;
                        v[j] = x;
                        const float a = fabsf(x);
                        mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                        mn1 = fminf(mn1, a);
                        sacc ^= __float_as_uint(x);
                    }
                    cst1 = mn2 * p.alpha;
                    cst2 = mn1 * p.alpha;
                }
                if (maxlvl > 0) __syncthreads();      // every read of the layer precedes its writes
                uint32_t pkn = 0u, idxn = 0u;
                if (act) {
                    // ---- pass 2: new c->v ; posterior = v->c + new c->v (primary edges)
#pragma unroll
                    for (int j = 0; j < DEG; j++) {
                        const float x = v[j];
                        const bool ismin = fabsf(x) == mn1;
                        const float mag = ismin ? cst1 : cst2;
                        const uint32_t s = (sacc ^ __float_as_uint(x)) & 0x80000000u;
                        const float nw = __uint_as_float(__float_as_uint(mag) | s);
                        pkn |= s >> (31 - j);
                        idxn = ismin ? (uint32_t)j : idxn;
                        if (!ent_absent(E[j], t) && ent_lvl(E[j]) == 0)
                            post_st<HYBRID>(E[j], ent_off(E[j], t), lpost, gpost, x + nw);
                    }
                    pkn |= idxn << 27;
                    C2V_ST(0, ci, cst1); C2V_ST(1, ci, cst2); C2V_ST(2, ci, __uint_as_float(pkn));
                    if (q == 1) { nx1 = cst1; nx2 = cst2; nxk = __uint_as_float(pkn); }
                }
                // ---- duplicate edges of a bit-group inside this layer: ordered delta updates
                for (int lvl = 1; lvl <= maxlvl; lvl++) {
                    __syncthreads();
                    if (act) {
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            if (ent_lvl(E[j]) == lvl && !(E[j] & LE_NULL)) {
                                const int off = ent_off(E[j], t);
                                const float nw = c2v_unpack(cst1, cst2, pkn, j);
                                const float od = c2v_unpack(c1o, c2o, pko, j);
                                const float L = post_ld<HYBRID>(E[j], off, lpost, gpost);
                                post_st<HYBRID>(E[j], off, lpost, gpost, L + (nw - od));
                            }
                        }
                    }
                }
                __syncthreads();
            }
            it++;
            if (p.early_stop || it == p.n_ite) {
                // ---- syndrome of the hard decisions (enable_syndrome, depth 1)
                int bad = 0;
                if (act)
                    for (int r = 0; r < q; r++) {
                        uint32_t x = 0u;
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            const LdpcEntry e = entries[r * p.ent_stride + j];
                            const float L = post_ld<HYBRID>(e, ent_off(e, t), lpost, gpost);
                            x ^= (!ent_absent(e, t) && L < 0.f) ? 1u : 0u;
                        }
                        bad |= (int)x;
                    }
                ok = !__syncthreads_or(bad);
                if (ok) break;
            }
        }

        // ---- outputs
        if (t == 0) {
            if (p.cwd) p.cwd[f] = ok ? 1 : 0;
            if (p.ites) p.ites[f] = it;
        }
        if (act) {
            for (int g = 0; g < p.n_info; g++) {
                const LdpcGroup gl = group_ld(groups, g);
                const float L = (HYBRID && !gl.lds) ? gpost[gl.base + t] : lpost[gl.base + t];
                if (p.bits) p.bits[(size_t)f * p.K + g * LDPC_Z + t] = L < 0.f ? 1 : 0;
                if (p.post) p.post[(size_t)f * p.N + g * LDPC_Z + t] = L;
            }
            if (p.post)
                for (int g = p.n_info; g < p.n_groups; g++) {
                    const LdpcGroup gl = group_ld(groups, g);
                    const float L = (HYBRID && !gl.lds) ? gpost[gl.base + t] : lpost[gl.base + t];
                    p.post[(size_t)f * p.N + p.K + q * t + (g - p.n_info)] = L;
                }
        }
        if (p.packed) {
            // bit i of word w = info bit 32 w + i (tail bits zero); 360 = 11.25 words per group, so pack by word
            const int n_words = (p.K + 31) / 32;
            for (int w = t; w < n_words; w += LDPC_THREADS) {
                uint32_t word = 0u;
                for (int b = 0; b < 32; b++) {
                    const int k = 32 * w + b;
                    if (k >= p.K) break;
                    const int g = k / LDPC_Z, m = k - g * LDPC_Z;
                    const LdpcGroup gl = group_ld(groups, g);
                    const float L = (HYBRID && !gl.lds) ? gpost[gl.base + m] : lpost[gl.base + m];
                    word |= (L < 0.f ? 1u : 0u) << b;
                }
                p.packed[(size_t)f * n_words + w] = word;
            }
        }
        __syncthreads();     // LDS is reused by the next frame of this workgroup
    }
#undef C2V_LD
#undef C2V_ST
}

template <int DEG, bool HYBRID, bool C2V_LDS>
static hipError_t launch_inst(const LdpcPlan &pl, const LdpcKParams &p, hipStream_t s)
{
    auto kern = ldpc_layered_nms_kernel<DEG, HYBRID, C2V_LDS>;
    static size_t configured_dev[64] = {0};
    int dev__ = 0;
    (void)hipGetDevice(&dev__);
    size_t &configured = configured_dev[dev__ & 63];
    if (pl.lds_bytes > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
        if (e != hipSuccess) return e;
        configured = pl.lds_bytes;
    }
    const int grid = p.n_frames < pl.grid_max ? p.n_frames : pl.grid_max;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LDPC_THREADS), pl.lds_bytes, s, p);
    return hipGetLastError();
}

// resident workgroups per CU for the instantiation the plan selects (persistent grid size)
template <int DEG, bool HYBRID, bool C2V_LDS>
static int occ_inst(const LdpcPlan &pl)
{
    auto kern = ldpc_layered_nms_kernel<DEG, HYBRID, C2V_LDS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, LDPC_THREADS, pl.lds_bytes) != hipSuccess) nb = 1;
    return nb < 1 ? 1 : nb;
}
int ldpc_blocks_per_cu(const LdpcPlan &pl)
{
    if (pl.fast && pl.fast_cu1) return 1;
    if (pl.fast && pl.fast_wg8) return ldpc_wg8_blocks_per_cu(pl);
    const bool small = pl.ent_stride == 13;
#define OCC(H, C) (small ? occ_inst<13, H, C>(pl) : occ_inst<LDPC_MAX_SLOTS, H, C>(pl))
    if (pl.hybrid) return pl.c2v_lds ? OCC(true, true) : OCC(true, false);
    return pl.c2v_lds ? OCC(false, true) : OCC(false, false);
#undef OCC
}

hipError_t ldpc_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s)
{
    if (pl.fast && pl.fast_cu1) return ldpc_cu1_launch(pl, p, s);
    if (pl.fast && pl.fast_wg8) return ldpc_wg8_launch(pl, p, s);
    p.entries = pl.d_entries; p.layer_deg = pl.d_layer_deg; p.layer_lvl = pl.d_layer_lvl; p.groups = pl.d_groups;
    p.N = pl.N; p.K = pl.K; p.M = pl.M; p.q = pl.q; p.n_info = pl.n_info; p.n_groups = pl.n_groups;
    p.ent_stride = pl.ent_stride; p.lds_post_words = pl.lds_post_words; p.glb_post_words = pl.glb_post_words;
    p.gwork_words = pl.gwork_words;
    const bool small = pl.ent_stride == 13;
#define DISPATCH(H, C)                                                      \
    (small ? launch_inst<13, H, C>(pl, p, s) : launch_inst<LDPC_MAX_SLOTS, H, C>(pl, p, s))
    if (pl.hybrid) return pl.c2v_lds ? DISPATCH(true, true) : DISPATCH(true, false);
    return pl.c2v_lds ? DISPATCH(false, true) : DISPATCH(false, false);
#undef DISPATCH
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------------
// (round 6; OPT-IN, DVBS2HIP_LDPC_ORDER=1: measured a 0.4 - 5 % LOSS, see dvbs2hip_api.hip ldpc_dev) Order in which the persistent grid's work queue hands out the frames
// of a launch with the stopping rule: noisiest first.  The idea: a frame that runs to the iteration cap takes 5 - 10 times the average, and one that starts last holds a
// workgroup while the chip idles; the mean |LLR| of a frame predicts those frames (tests/test_ldpc_gpu.py: 90 % of the noisier half do not converge).  The measurement:
// near the waterfall the AVERAGE frame already takes ~10 iterations and the counter-fed queue balances the rest -- there is no tail to hide.  Two small kernels in front of the
// decoder: sum |LLR| per frame (one streaming pass, ~0.1 ms per 8192 short frames), then a counting sort of the frames into 1024 buckets of that sum (one workgroup).
// Results do not depend on it: every frame is decoded exactly once, into its own sockets.
__global__ void __launch_bounds__(256)
frame_metric_kernel(const float *llr, float *metric, int N)
{
    const float *x = llr + (size_t)blockIdx.x * N;
    float a = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) a += fabsf(x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) metric[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ void __launch_bounds__(1024)
frame_order_kernel(const float *metric, uint32_t *order, int F)
{
    __shared__ uint32_t hist[1024], base[1024];
    __shared__ float red[2][16];
    const int t = threadIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = t; i < F; i += 1024) { const float m = metric[i]; if (m == m && m < INFINITY) { lo = fminf(lo, m); hi = fmaxf(hi, m); } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if ((t & 63) == 0) { red[0][t >> 6] = lo; red[1][t >> 6] = hi; }
    hist[t] = 0u;
    __syncthreads();
    lo = red[0][0]; hi = red[1][0];
    for (int k = 1; k < 16; k++) { lo = fminf(lo, red[0][k]); hi = fmaxf(hi, red[1][k]); }
    const float scale = hi > lo ? 1023.0f / (hi - lo) : 0.f;
    auto bucket = [&](float m) -> uint32_t { const float b = (m - lo) * scale; return b >= 0.f ? (b < 1023.f ? (uint32_t)b : 1023u) : 0u; };      // (NaN -> bucket 0: any bucket keeps `order` a permutation)
    for (int i = t; i < F; i += 1024) atomicAdd(&hist[bucket(metric[i])], 1u);
    __syncthreads();
    // exclusive prefix sum of the 1024 counts (Hillis-Steele in LDS)
    base[t] = hist[t];
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = t >= o ? base[t - o] : 0u;
        __syncthreads();
        base[t] += v;
        __syncthreads();
    }
    const uint32_t excl = base[t] - hist[t];
    __syncthreads();
    base[t] = excl;
    __syncthreads();
    for (int i = t; i < F; i += 1024) order[atomicAdd(&base[bucket(metric[i])], 1u)] = (uint32_t)i;      // smallest sums (the noisiest frames) first
}

hipError_t frame_order_launch(const float *llr, float *metric, uint32_t *order, int F, int N, hipStream_t s)
{
    hipLaunchKernelGGL(frame_metric_kernel, dim3(F), dim3(256), 0, s, llr, metric, N);
    hipLaunchKernelGGL(frame_order_kernel, dim3(1), dim3(1024), 0, s, (const float *)metric, order, F);
    return hipGetLastError();
}

}  // namespace dvbs2
