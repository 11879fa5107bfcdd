// gen_rows.h — the two-stage row kernel of the general-size sweep (gen_rows.hip), as seen by general_host.inc.
//
// k_gen_rows<R1, R2> handles rows of nx = R1 * R2 points (R1, R2 <= 32): both radix steps run in registers, the row
// crosses LDS once between them.  Sizes without such a factorisation keep k_gen_fused (general_sizes.inc).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#define HH_HIDDEN __attribute__((visibility("hidden")))

struct GenRowsArgs {
  const float2* table;    // [runs][cap][nky] run tables (k_gen_run_table)
  const int* run_imax;    // [runs]
  const int* layer_run;   // [layers] grid layer -> run
  const int* layer_first; // [layers] first candidate of the layer inside the batch
  const int* layer_count; // [layers] candidates of the layer
  const float* eg;        // [B][kg][nxp] column factors (k_gen_column_factors)
  const int* cgs;         // [B][nxp/4 + 4] first table row per column group, then the candidate's row count
  const float2* w2;       // [nky][nx] {w, w (E - Ebar)}
  const float2* tw_nx;    // [nx] exp(-2 pi i k / nx)
  double* partials;       // [B][nky][3]
  float* q_out;           // several segments: [B][q_stride] masked q, bin (ky, kx) at ky nx + kx (NULL: one segment)
  int64_t q_stride;
  int cap, rows_lds, kg, n_units, log_flag;
  int nky, nx, nxp;
  int halves;             // 2: the column factors are double-buffered in LDS (set by gen_rows_launch from the plan)
};

struct GenRowsPlan {
  int r1, r2;             // nx = r1 * r2; 0 when the size has no supported factorisation
  int rows_per_block;     // spectrum rows (ky) one workgroup carries through a layer's candidates
  int threads;
  int halves;             // LDS buffers for the column factors (2 = double-buffered, one barrier per candidate)
  size_t lds;             // dynamic LDS bytes for (rows_lds, kg)
};

// one instantiated kernel (the four parts of gen_rows.hip each hold a share of the list)
struct GenRowsEntry {
  int r1, r2;
  void (*kernel)(GenRowsArgs);
  int rpb, row_len, nxp, pf, tw_lds, max_blocks;
};
HH_HIDDEN const GenRowsEntry* gen_rows_part_0(int* n);
HH_HIDDEN const GenRowsEntry* gen_rows_part_1(int* n);
HH_HIDDEN const GenRowsEntry* gen_rows_part_2(int* n);
HH_HIDDEN const GenRowsEntry* gen_rows_part_3(int* n);

// the factorisation for nx (r1 = 0: none), then the launch shape for a batch's (rows_lds, kg); false when it does not
// fit (LDS, or more column factors than the register prefetch holds)
HH_HIDDEN bool gen_rows_plan(int nx, int rows_lds, int kg, GenRowsPlan* plan);
HH_HIDDEN hipError_t gen_rows_prepare(const GenRowsPlan& plan, int* blocks_per_cu);   // LDS attribute + occupancy
HH_HIDDEN hipError_t gen_rows_launch(const GenRowsPlan& plan, int n_ky_blocks, int layers, hipStream_t stream, const GenRowsArgs& args);
