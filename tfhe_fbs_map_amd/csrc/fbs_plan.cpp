// plan_program: see fbs_plan.hpp.  Host only; no HIP header is included here.
#include "fbs_plan.hpp"

#include <algorithm>
#include <functional>

namespace fbs {

int plan_program(const fbs_program_desc *d, uint32_t n_tables, const uint8_t *fusable, ProgramPlan *out, std::string *err) {
    auto fail = [&](const std::string &msg) {
        if (err) *err = msg;
        return FBS_E_INVALID;
    };
    if (!d || !out) return fail("null argument");
    // counts are checked BEFORE anything is sized by them (wire ids are 32-bit: n_inputs + n_instr must not wrap)
    if ((uint64_t)d->n_inputs + d->n_instr > FBS_MAX_WIRES || d->n_terms > FBS_MAX_TERMS || d->n_outputs > FBS_MAX_WIRES)
        return fail("program too large: n_inputs + n_instr and n_outputs are bounded by FBS_MAX_WIRES, n_terms by FBS_MAX_TERMS");
    if ((d->n_instr && (!d->kind || !d->arg0 || !d->arg1 || !d->const_coef)) || (d->n_terms && (!d->term_coef || !d->term_src)) ||
        (d->n_outputs && !d->out_wire))
        return fail("null array in the program description");
    ProgramPlan &plan = *out;
    plan = ProgramPlan();
    const uint32_t n_wires = plan.n_wires = d->n_inputs + d->n_instr;

    // levels: inputs 0, LinearProd = max over sources, Bootstrap = source + 1
    std::vector<uint32_t> level(n_wires, 0), sub(n_wires, 0);
    std::vector<uint8_t> is_lin(n_wires, 0);
    for (uint32_t i = 0; i < d->n_instr; i++) {
        const uint32_t w = d->n_inputs + i;
        if (d->kind[i] == 0) {
            is_lin[w] = 1;
            if ((uint64_t)d->arg0[i] + d->arg1[i] > d->n_terms) return fail("term range out of bounds");
            uint32_t lv = 0, sb = 0;
            for (uint32_t t = d->arg0[i]; t < d->arg0[i] + d->arg1[i]; t++) {
                const uint32_t s = d->term_src[t];
                if (s >= w) return fail("instruction " + std::to_string(i) + " reads a later wire");
                lv = std::max(lv, level[s]);
            }
            for (uint32_t t = d->arg0[i]; t < d->arg0[i] + d->arg1[i]; t++) {
                const uint32_t s = d->term_src[t];
                if (is_lin[s] && level[s] == lv) sb = std::max(sb, sub[s] + 1);
            }
            level[w] = lv;
            sub[w] = sb;
        } else if (d->kind[i] == 1) {
            if (d->arg0[i] >= w) return fail("instruction " + std::to_string(i) + " reads a later wire");
            if (d->arg1[i] >= n_tables) return fail("table id out of range");
            level[w] = level[d->arg0[i]] + 1;
            plan.depth = std::max(plan.depth, level[w]);
            plan.n_bootstrap++;
        } else {
            return fail("unknown instruction kind");
        }
    }
    for (uint32_t o = 0; o < d->n_outputs; o++)
        if (d->out_wire[o] >= (int64_t)n_wires) return fail("output wire out of range");

    const uint32_t depth = plan.depth;
    // ---- stages in execution order: for each level its lincomb sub-stages, then its bootstraps ----------------
    std::vector<uint32_t> n_sub(depth + 1, 0);
    for (uint32_t w = d->n_inputs; w < n_wires; w++)
        if (is_lin[w]) n_sub[level[w]] = std::max(n_sub[level[w]], sub[w] + 1);
    std::vector<uint32_t> lin_time0(depth + 1, 0), boot_time(depth, 0);
    uint32_t n_stages = 0;
    for (uint32_t L = 0; L <= depth; L++) {
        lin_time0[L] = n_stages;
        n_stages += n_sub[L];
        if (L < depth) boot_time[L] = n_stages++;
    }
    auto def_time = [&](uint32_t w) -> int64_t {
        if (w < d->n_inputs) return -1;
        return is_lin[w] ? (int64_t)lin_time0[level[w]] + sub[w] : (int64_t)boot_time[level[w] - 1];
    };
    // ---- liveness: a wire keeps its slot until the stage of its last reader has run (outputs: for ever) ---------
    const int64_t FOREVER = (int64_t)n_stages + 1;
    std::vector<int64_t> last(n_wires);
    for (uint32_t w = 0; w < n_wires; w++) last[w] = def_time(w);
    for (uint32_t i = 0; i < d->n_instr; i++) {
        const uint32_t w = d->n_inputs + i;
        const int64_t t = def_time(w);
        if (d->kind[i] == 0) {
            for (uint32_t k = d->arg0[i]; k < d->arg0[i] + d->arg1[i]; k++) last[d->term_src[k]] = std::max(last[d->term_src[k]], t);
        } else {
            last[d->arg0[i]] = std::max(last[d->arg0[i]], t);
        }
    }
    for (uint32_t o = 0; o < d->n_outputs; o++)
        if (d->out_wire[o] >= 0) last[d->out_wire[o]] = FOREVER;
    std::vector<std::vector<uint32_t>> born(n_stages + 1), dies(n_stages + 1);   // index = time + 1
    for (uint32_t w = 0; w < n_wires; w++) {
        born[def_time(w) + 1].push_back(w);
        if (last[w] != FOREVER) dies[last[w] + 1].push_back(w);
    }
    std::vector<uint32_t> slot(n_wires, 0), free_slots;   // free_slots: min-heap, lowest slot first (compact buffer)
    auto cmp = std::greater<uint32_t>();
    uint32_t n_slots = 0;
    for (uint32_t t = 0; t <= n_stages; t++) {
        for (uint32_t w : born[t]) {
            if (free_slots.empty()) {
                slot[w] = n_slots++;
            } else {
                std::pop_heap(free_slots.begin(), free_slots.end(), cmp);
                slot[w] = free_slots.back();
                free_slots.pop_back();
            }
        }
        // freed only now: a slot is never rewritten by the stage that reads it last
        for (uint32_t w : dies[t]) {
            free_slots.push_back(slot[w]);
            std::push_heap(free_slots.begin(), free_slots.end(), cmp);
        }
    }
    plan.n_slots = std::max(1u, n_slots);
    plan.in_slot.assign(slot.begin(), slot.begin() + d->n_inputs);
    plan.out_slot.resize(d->n_outputs);
    for (uint32_t o = 0; o < d->n_outputs; o++) plan.out_slot[o] = d->out_wire[o] >= 0 ? (int64_t)slot[d->out_wire[o]] : d->out_wire[o];

    // ---- stage tables (wire slots, not wire ids) ---------------------------------------------------------------
    plan.lin.assign(depth + 1, {});
    plan.boot.assign(depth, {});
    struct Gate {
        uint32_t src, dst, tab;
    };
    std::vector<std::vector<Gate>> bh(depth);
    for (uint32_t i = 0; i < d->n_instr; i++) {
        const uint32_t w = d->n_inputs + i;
        if (d->kind[i] == 0) {
            auto &stages = plan.lin[level[w]];
            if (stages.size() <= sub[w]) stages.resize(sub[w] + 1);   // (kept even when empty: stage times above count every sub-stage)
            LinPlan &h = stages[sub[w]];
            h.dst.push_back(slot[w]);
            for (uint32_t t = d->arg0[i]; t < d->arg0[i] + d->arg1[i]; t++) {
                h.srcs.push_back(slot[d->term_src[t]]);
                h.coefs.push_back(d->term_coef[t]);
            }
            h.off.push_back((uint32_t)h.srcs.size());
            h.consts.push_back(d->const_coef[i]);
        } else {
            bh[level[w] - 1].push_back({d->arg0[i], slot[w], d->arg1[i]});
        }
    }
    for (uint32_t L = 0; L < depth; L++) {
        std::vector<Gate> &gates = bh[L];
        std::stable_sort(gates.begin(), gates.end(), [](const Gate &x, const Gate &y) { return x.src < y.src; });
        BootPlan &st = plan.boot[L];
        for (size_t g = 0; g < gates.size();) {
            size_t e = g;
            while (e < gates.size() && gates[e].src == gates[g].src) e++;
            st.src_slot.push_back(slot[gates[g].src]);
            const uint32_t u = (uint32_t)st.src_slot.size() - 1;
            // fused: the tables of this source that k_multi_extract can serve share one rotation of TV_0, if there are
            // at least two of them; the others keep a rotation of their own
            size_t n_fusable = 0;
            if (fusable)
                for (size_t i = g; i < e; i++) n_fusable += fusable[gates[i].tab];
            const bool share = n_fusable >= 2;
            const uint32_t shared_at = (uint32_t)st.dst.size();
            if (share) {
                st.source_of.push_back(u);
                st.dst.push_back(0x80000000u | st.n_shared);
                st.table.push_back(n_tables);
            }
            for (size_t i = g; i < e; i++) {
                if (share && fusable[gates[i].tab]) {
                    st.x_row.push_back(st.n_shared);
                    st.x_gate.push_back(shared_at);
                    st.x_table.push_back(gates[i].tab);
                    st.x_dst.push_back(gates[i].dst);
                } else {
                    st.source_of.push_back(u);
                    st.dst.push_back(gates[i].dst);
                    st.table.push_back(gates[i].tab);
                }
            }
            if (share) st.n_shared++;
            g = e;
        }
        plan.max_width = std::max(plan.max_width, (uint32_t)st.dst.size());
        plan.max_sources = std::max(plan.max_sources, (uint32_t)st.src_slot.size());
        plan.max_shared = std::max(plan.max_shared, st.n_shared);
        plan.n_keyswitch += (uint32_t)st.src_slot.size();
        plan.n_rotations += (uint32_t)st.dst.size();
    }
    return FBS_OK;
}

}  // namespace fbs
