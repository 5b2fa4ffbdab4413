// No packed-fp32 instructions (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) in any kernel of this library.
//
// On MI355X a chain of dependent v_pk_*_f32 instructions with op_sel modifiers was found to return a wrong LOW half for lanes
// 16..31 / 48..63 about once per 2e6 executions -- the value short of exactly one term of the chain -- depending on the alignment
// of the code (period 32 bytes) and only with at least two waves per SIMD; the same arithmetic as v_fma_f32 never fails
// (tools/probes/pk_chain_probe_pad.hip + run_pad_sweep.sh: the stand-alone reproducer; profiles/r05_mol_fused2_soak.txt: how it
// was found in molfuse2.hip).  Which kernels are exposed changes with every recompile, so the instruction class is switched off
// for all of them; measured cost: none (profiles/r05_no_packed_fp32_ab.txt).  Every .hip file includes this first and
// nopk_end.h last; device pass only -- the host pass does not know the feature.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NNHIP_PACKED_FP32)
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#endif
