// Template-specialised batch solver: the header of a template's plan block (wave_plan.h builds the block, wave_ipm.h /
// the kernels read it).  Free of standard-library includes: this text also travels inside the library
// (wave_hdr_src.inc) into the per-template kernels of wave_codegen.h.
#pragma once
#include "exec.h"

namespace dnlp {

// outputs of a product by output (tape.h CooIdx) with more entries than this are summed by all lanes of a wavefront
constexpr int kCooHeavy = 12;

// header of the block (all 32-bit; offsets count ints from the start of the block)
struct WaveHdr {
  i32 total;                                  // ints in the block
  i32 N, m, Z, nd, nh, nnzJ, nnzH, nunits;
  i32 u_op, u_a0, u_a1, u_z, u_d0, u_d1, u_h, u_p;          // per sweep unit (see build_wave_plan)
  i32 mm_idx;                                 // matmul units: (U entry, V entry) index pairs of every inner product
  i32 keep_gen;                               // ints of the block a kernel with generated LDL^T phases (wave_gen.h) still reads: the tables
                                              // of the level machinery come last and are not staged (a template without dense tail also drops fa / fu0 / fu1)
  i32 G_ptr, G_idx, Mg_ptr, Mg_idx, MJ_ptr, MJ_idx, Mw_ptr, Mw_idx, MH_ptr, MH_idx;
  i32 jac_rows, jac_cols, hess_rows, hess_cols, jac_rowptr;
  // products by output: J v (rows), J^T v (columns), sym(H) v
  i32 jr_ptr, jr_ent, jr_src, jr_heavy, jr_nheavy;
  i32 jc_ptr, jc_ent, jc_src, jc_heavy, jc_nheavy;
  i32 hs_ptr, hs_ent, hs_src, hs_heavy, hs_nheavy;
  // static-pattern LDL^T
  i32 sp_nblk, sp_nvals, sp_nlev, sp_ngrp, sp_nfwd, sp_ntrip, sp_rows;
  i32 bnode, soff, loff, doff, lev_off, sblk, sidx, lev_f, fnode, foff, fa, fu0, fu1, lev_g, gdst, goff, tau, tav, hpos, jpos, dpos;
  i32 lev_r, lev_t, lev_fe, keep_pad;          // per level (nlev + 1 each): first struct row, first update triple, first gathered row
  // DENSE TAIL: the last tail_T levels are a chain of one 1x1 block each over a dense trailing matrix (a dense separator:
  // circle packing n = 10 ends in 21 such levels of 23).  wave_ipm.h factors and solves that matrix in registers — one
  // row per lane — instead of walking the level machinery once per block.  tail_L = first tail level (= nlev: no tail).
  i32 tail_L, tail_T;
  i32 t_node, t_d, t_l;                       // tail position -> KKT node / value index of D; (i, j), i > j -> value index of L_ij (T x T)
  i32 t_fq, t_fp;                             // forward gathers of the tail targets from blocks BEFORE the tail: entry list, T + 1 offsets into it
  i32 t_nf;                                   // entries in t_fq
  // data row of an instance (doubles; batch.h BatchLayout)
  i32 l_c0, l_c, l_b, l_Jc, l_G, l_Mg, l_Mw, l_MJ, l_MH, l_fp, l_fp2, l_x0, l_lb, l_ub, l_cl, l_cu, l_total;
  i32 state_doubles;                          // solver state of one instance (wave_ipm.h layout)
  i32 scr_doubles;                            // its scratch array: the largest phase of products (wave_ipm.h run_sum)
};
static_assert(sizeof(WaveHdr) % 8 == 0, "the tables behind the header start 8-byte aligned");

}  // namespace dnlp
