// The schedule of a program, computed on the host and without a device: levels, wire slots by liveness, the stage tables the
// level kernels index (the loop of the reference's `LutExecEnv.eval`, fbs_mapper/fbs_exec_env.py:208-229, turned into
// level-batched stages).  Plain C++ on the plain-C program description of include/fbs_exec.h: fbs_program_load_ex (fbs_capi.cpp)
// uploads what `plan_program` returns, and the sanitizer harness of tests/c/ runs the same function -- and executes its result in
// the clear -- under AddressSanitizer / UBSan with no GPU in sight.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/fbs_exec.h"

namespace fbs {

// linear combinations of one sub-stage of a level (outputs that depend on each other sit in consecutive sub-stages)
struct LinPlan {
    std::vector<uint32_t> dst, off{0}, srcs;   // wire SLOTS; off[o] .. off[o + 1]: the terms of output o
    std::vector<int64_t> coefs, consts;        // as in the description (the caller maps them into the field)
};
// Bootstraps of one level, sorted by source wire.  Gates that read the same wire (the reference's one-gate-one-bootstrap
// lowering emits several tables per linear combination, fbs_mapper/map_to_fbs.py:41-45, and its CSE only merges identical
// tables, fbs_mapper/fbs_exec_env.py:93-100) share one key switch + modulus switch.
struct BootPlan {
    std::vector<uint32_t> src_slot;            // [n_sources] wire slot of each distinct source
    std::vector<uint32_t> source_of, dst, table;   // [n_gates] index into src_slot, wire slot (or 0x80000000 | shared index), table id
    // fused plans: the tables of a source that several read are served by ONE gate of the list above -- the rotation of TV_0
    // (table id = n_tables, dst = 0x80000000 | shared index) -- and one entry each of the extraction list below
    uint32_t n_shared = 0;
    std::vector<uint32_t> x_row, x_table, x_dst, x_gate;   // [n_extract] shared index, table, wire slot, position of the rotation in the gate list
};
struct ProgramPlan {
    uint32_t n_wires = 0, n_slots = 0, depth = 0, max_width = 0, max_sources = 0, max_shared = 0;
    uint32_t n_bootstrap = 0, n_keyswitch = 0, n_rotations = 0;
    std::vector<uint32_t> in_slot;             // [n_inputs]
    std::vector<int64_t> out_slot;             // [n_outputs]  slot, or -1-c for the constant c
    std::vector<std::vector<LinPlan>> lin;     // [depth + 1][sub-stage]
    std::vector<BootPlan> boot;                // [depth]  (boot[L] = bootstraps of level L + 1)
};

// FBS_OK, or FBS_E_INVALID with *err.  `fusable`: null = one blind rotation per table; else [n_tables] flags of the tables whose
// D_F is small enough for k_multi_extract (FBS_LOAD_FUSE_TABLES: two or more such tables on one source share a rotation of TV_0).
int plan_program(const fbs_program_desc *d, uint32_t n_tables, const uint8_t *fusable, ProgramPlan *out, std::string *err);

}  // namespace fbs
