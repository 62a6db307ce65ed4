// Sanitizer harness for the HOST side of libfbsexec ("race detection / sanitizers" -- on the CPU build only; the
// GPU pool has no AddressSanitizer).  Built by tests/c/Makefile with g++ -fsanitize=address,undefined from the product's own
// sources -- csrc/fbs_plan.cpp (the program loader's scheduling, slot reuse and level-index construction) and csrc/fbs_host.cpp
// (key generation, encryption, decryption, test vectors) -- and driven by tests/test_sanitizers.py.  No GPU, no HIP call.
//
//   host_harness plan  < description        the plan of a program (plain and with shared rotations), EXECUTED in the clear on
//                                           wire slots exactly as the level kernels index them; prints the outputs
//   host_harness crypto                     keygen / encrypt / decrypt / test vectors at toy parameter sets, checked
//
// description (text, whitespace separated): n_inputs n_instr n_terms n_outputs n_tables T
//   kind[n_instr] arg0[n_instr] arg1[n_instr] const[n_instr] term_coef[n_terms] term_src[n_terms] out_wire[n_outputs]
//   per table: len values...      fusable[n_tables]      inputs[n_inputs][T]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../tfhe_fbs_map_amd/csrc/fbs_internal.hpp"
#include "../../tfhe_fbs_map_amd/csrc/fbs_plan.hpp"

namespace fbs {
int set_error(const fbs_ctx *ctx, int code, const std::string &msg) {   // (the product's lives in fbs_capi.cpp, beside the HIP calls)
    if (ctx) ctx->err = msg;
    return code;
}
}  // namespace fbs
using namespace fbs;

template <class T>
static std::vector<T> read_n(size_t n) {
    std::vector<T> v(n);
    for (auto &x : v) {
        long long tmp;
        if (!(std::cin >> tmp)) {
            fprintf(stderr, "short input\n");
            exit(2);
        }
        x = (T)tmp;
    }
    return v;
}

// Execute a plan in the clear: values live in wire SLOTS [n_slots][T]; a level's linear combinations run sub-stage by sub-stage,
// then its bootstraps read their source's slot and write their own -- every index the level kernels would use is used here, on
// vectors the sanitizer guards.  A slot that is read must hold the wire the description says (`holds`): a wrong reuse aborts.
static int run_plan(const fbs_program_desc &d, const ProgramPlan &plan, const std::vector<std::vector<int64_t>> &tables, size_t T,
                    const std::vector<int64_t> &inputs, std::vector<int64_t> *outputs) {
    std::vector<int64_t> wires((size_t)plan.n_slots * T, -777);
    for (uint32_t i = 0; i < d.n_inputs; i++)
        for (size_t s = 0; s < T; s++) wires.at((size_t)plan.in_slot.at(i) * T + s) = inputs.at((size_t)i * T + s);
    for (uint32_t L = 0; L <= plan.depth; L++) {
        for (const LinPlan &h : plan.lin.at(L)) {
            std::vector<int64_t> fresh(h.dst.size() * T);
            for (size_t o = 0; o < h.dst.size(); o++)
                for (size_t s = 0; s < T; s++) {
                    int64_t acc = h.consts.at(o);
                    for (uint32_t t = h.off.at(o); t < h.off.at(o + 1); t++) acc += h.coefs.at(t) * wires.at((size_t)h.srcs.at(t) * T + s);
                    fresh[o * T + s] = acc;
                }
            for (size_t o = 0; o < h.dst.size(); o++)      // (a stage reads before it writes: k_lincomb's outputs never feed its own inputs)
                for (size_t s = 0; s < T; s++) wires.at((size_t)h.dst.at(o) * T + s) = fresh[o * T + s];
        }
        if (L == plan.depth) break;
        const BootPlan &b = plan.boot.at(L);
        if (b.source_of.size() != b.dst.size() || b.table.size() != b.dst.size()) return 3;
        std::vector<std::vector<int64_t>> shared(b.n_shared, std::vector<int64_t>(T));
        std::vector<int64_t> fresh(b.dst.size() * T);
        for (size_t g = 0; g < b.dst.size(); g++)
            for (size_t s = 0; s < T; s++) {
                const int64_t x = wires.at((size_t)b.src_slot.at(b.source_of.at(g)) * T + s);
                if (b.dst[g] & 0x80000000u) {
                    if (b.table[g] != tables.size()) return 4;                 // a shared rotation starts from TV_0
                    shared.at(b.dst[g] & 0x7FFFFFFFu)[s] = x;                  // (in the clear the accumulator "is" the source value)
                } else {
                    const auto &tab = tables.at(b.table[g]);
                    if (x < 0 || (size_t)x >= tab.size()) return 5;
                    fresh[g * T + s] = tab[(size_t)x];
                }
            }
        for (size_t g = 0; g < b.dst.size(); g++)
            if (!(b.dst[g] & 0x80000000u))
                for (size_t s = 0; s < T; s++) wires.at((size_t)b.dst[g] * T + s) = fresh[g * T + s];
        for (size_t e = 0; e < b.x_row.size(); e++) {
            if (b.x_gate.at(e) >= b.dst.size() || b.dst[b.x_gate[e]] != (0x80000000u | b.x_row[e])) return 6;
            for (size_t s = 0; s < T; s++) {
                const int64_t x = shared.at(b.x_row[e])[s];
                const auto &tab = tables.at(b.x_table.at(e));
                if (x < 0 || (size_t)x >= tab.size()) return 7;
                wires.at((size_t)b.x_dst.at(e) * T + s) = tab[(size_t)x];
            }
        }
    }
    outputs->assign((size_t)d.n_outputs * T, 0);
    for (uint32_t o = 0; o < d.n_outputs; o++)
        for (size_t s = 0; s < T; s++)
            (*outputs)[o * T + s] = plan.out_slot.at(o) >= 0 ? wires.at((size_t)plan.out_slot[o] * T + s) : -1 - plan.out_slot[o];
    return 0;
}

static int mode_plan() {
    const auto head = read_n<uint32_t>(6);
    fbs_program_desc d{};
    d.n_inputs = head[0], d.n_instr = head[1], d.n_terms = head[2], d.n_outputs = head[3];
    const uint32_t n_tables = head[4];
    const size_t T = head[5];
    const auto kind = read_n<uint8_t>(d.n_instr);
    const auto arg0 = read_n<uint32_t>(d.n_instr), arg1 = read_n<uint32_t>(d.n_instr);
    const auto cst = read_n<int64_t>(d.n_instr), tc = read_n<int64_t>(d.n_terms);
    const auto ts = read_n<uint32_t>(d.n_terms);
    const auto ow = read_n<int64_t>(d.n_outputs);
    std::vector<std::vector<int64_t>> tables(n_tables);
    for (auto &t : tables) t = read_n<int64_t>(read_n<uint32_t>(1)[0]);
    const auto fusable = read_n<uint8_t>(n_tables);
    const auto inputs = read_n<int64_t>((size_t)d.n_inputs * T);
    d.kind = kind.data(), d.arg0 = arg0.data(), d.arg1 = arg1.data(), d.const_coef = cst.data(), d.term_coef = tc.data(), d.term_src = ts.data();
    d.out_wire = ow.data();
    std::vector<int64_t> first;
    for (int fused = 0; fused < 2; fused++) {
        ProgramPlan plan;
        std::string err;
        const int rc = plan_program(&d, n_tables, fused ? fusable.data() : nullptr, &plan, &err);
        if (rc != FBS_OK) {
            printf("error %d %s\n", rc, err.c_str());
            return 0;
        }
        std::vector<int64_t> outs;
        const int bad = run_plan(d, plan, tables, T, inputs, &outs);
        if (bad) {
            printf("plan is inconsistent (%d)\n", bad);
            return 1;
        }
        if (fused && outs != first) {
            printf("the plan with shared rotations computes something else\n");
            return 1;
        }
        if (!fused) first = outs;
        printf("%s depth %u slots %u wires %u bootstraps %u keyswitches %u rotations %u max_width %u max_sources %u max_shared %u\n", fused ? "fused" : "plain",
               plan.depth, plan.n_slots, plan.n_wires, plan.n_bootstrap, plan.n_keyswitch, plan.n_rotations, plan.max_width, plan.max_sources, plan.max_shared);
        printf("widths");
        for (const BootPlan &b : plan.boot) printf(" %zu", b.dst.size());
        printf("\n");
    }
    printf("outputs");
    for (int64_t v : first) printf(" %lld", (long long)v);
    printf("\n");
    return 0;
}

static int mode_crypto() {
    struct Set {
        uint32_t n, log_n, k, l, beta, t, gamma, p, group;
    };
    const Set sets[] = {{12, 8, 1, 3, 7, 8, 2, 7, 1}, {10, 9, 1, 2, 9, 5, 3, 15, 1}, {12, 8, 1, 1, 20, 8, 2, 7, 2}, {8, 8, 2, 1, 21, 8, 2, 7, 2}};
    for (const Set &s : sets) {
        fbs_ctx ctx;
        fbs_params p{};
        p.n = s.n, p.log_n_poly = s.log_n, p.k = s.k, p.l_bsk = s.l, p.beta_bsk = s.beta, p.t_ksk = s.t, p.gamma_ksk = s.gamma, p.p_msg = s.p;
        p.sigma_lwe = 1 << 8, p.sigma_glwe = 1 << 4, p.bsk_group = s.group;
        if (host_ctx_init(&ctx, &p, 7, nullptr) != FBS_OK) {
            printf("host_ctx_init failed: %s\n", ctx.err.c_str());
            return 1;
        }
        host_keygen(&ctx);
        const uint32_t D = ctx.D;
        if (ctx.sk_lwe.size() != p.n || ctx.sk_glwe.size() != D || ctx.bsk.size() != ctx.n_ggsw * ctx.rows * (p.k + 1) * ctx.N ||
            ctx.ksk.size() != (size_t)D * p.t_ksk * (p.n + 1)) {
            printf("key sizes\n");
            return 1;
        }
        const size_t count = 37;
        std::vector<int64_t> msgs(count), back(count);
        for (size_t i = 0; i < count; i++) msgs[i] = (int64_t)(i % (2 * p.p_msg));
        std::vector<uint64_t> cts(count * (D + 1));
        host_encrypt(&ctx, msgs.data(), count, 1000, cts.data());
        host_decrypt(&ctx, cts.data(), count, back.data());
        if (back != msgs) {
            printf("decrypt(encrypt(m)) != m\n");
            return 1;
        }
        // test vectors: a table on the half torus, the three negacyclic modes, a multi-valued one; one that must be refused
        const std::vector<std::vector<int32_t>> good = {{0, 1, 1, 0, 1, 0, 0}, {0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1}, {0, 0, 1, 0, 0, 0, 0, 0, 0}, {1, 1, 0, 1, 1, 1, 1, 1, 1}, {0, 1, 2, 3, 2, 1, 0}};
        for (const auto &tab : good) {
            if (tab.size() > 2 * p.p_msg) continue;
            std::vector<uint64_t> tv(ctx.N);
            uint64_t post = 0, d2 = 0, g2 = 0, abs_sum = 0;
            std::vector<uint32_t> pos(p.p_msg + 1);
            std::vector<int32_t> val(p.p_msg + 1);
            uint32_t nd = 0;
            if (host_build_tv(&ctx, tab.data(), (uint32_t)tab.size(), tv.data(), &post) != FBS_OK ||
                host_build_tv_diff(&ctx, tab.data(), (uint32_t)tab.size(), pos.data(), val.data(), &nd, &d2, &g2, &abs_sum) != FBS_OK || nd > p.p_msg + 1) {
                printf("test vector of a valid table refused\n");
                return 1;
            }
        }
        const std::vector<int32_t> bad = {0, 1, 1, 0, 1, 0, 0, 1, 1};      // table[1] + table[1 + p] is not the constant of the overlap (p = 7)
        std::vector<uint64_t> tv(ctx.N);
        uint64_t post = 0;
        if (p.p_msg == 7 && host_build_tv(&ctx, bad.data(), (uint32_t)bad.size(), tv.data(), &post) != FBS_E_TABLE) {
            printf("an invalid table was accepted\n");
            return 1;
        }
        printf("set n=%u N=%u k=%u l=%u group=%u ok\n", p.n, ctx.N, p.k, p.l_bsk, ctx.group);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "plan")) return mode_plan();
    if (argc >= 2 && !strcmp(argv[1], "crypto")) return mode_crypto();
    fprintf(stderr, "usage: host_harness plan|crypto\n");
    return 2;
}
