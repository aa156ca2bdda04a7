/* lookahead_hip_lab.h — the kernel LAB of liblookahead_hip.so: measurement knobs and A/B switches for the scripts under scripts/
 * and the bitwise-identity tests.  NOT part of the product boundary (include/lookahead_hip.h): nothing a reference maintainer
 * binds, no stability promise.  Every knob has a library default (la_lab_get reports the value in effect); the step entry points
 * capture their graphs again after a change.
 *
 * Defaults: everything 0 except key 6 = 12657 (paired wide launches + their fat-wave forms; gate/up and QKV unpaired at <= 4 blocks, QKV over <= 128 regions in two token groups), key 11 = 1 (one step per graph), key 17 = 1
 * (single-launch tree attention), key 24 = 1, key 25 = 13.
 */
#ifndef LOOKAHEAD_HIP_LAB_H
#define LOOKAHEAD_HIP_LAB_H
#ifdef __cplusplus
extern "C" {
#endif
/* Measurement knobs for the kernel A/B scripts (scripts/gpu_ab.py); every knob is 0 in production.
 * key 0: GEMM kernels return after the weight-streaming loop, before the cross-wave reduction and epilogue.
 * key 1: K share (1/64ths) of waves 0..3 in the 8-wave GEMMs (0 = library default); set before la_llama_step captures.
 * key 2: s_setprio level (0..3) of waves 4..7 in the 8-wave GEMMs.
 * key 3: 1 = multi-block GEMMs always on the K-split kernels (no wide one-pass form); set before la_llama_mstep captures.
 * key 6: paired form of the wide multi-block launches (two weight regions x half the token blocks per workgroup): bit 0 = slab and
 *        QKV launches (library default: on), bit 1 = gate/up too, bit 2 = quad QKV form, bit 3 = QKV with token quarters at every
 *        block count (what the second QKV image, cfg.qkv_mb_wg, takes by default), bit 4 (round 5) = gate/up as four fat waves per
 *        workgroup (k_gemm_fat: the paired geometry with 4 x TW accumulator tiles per wave, one wave per SIMD; library default: on),
 *        bit 5 = the paired slab / QKV launches as fat waves too (library default: on), bit 6 = gate/up at <= 4 blocks as ONE region x all token
 *        blocks per workgroup (fat waves, nt weights; default on), bit 7 = that form at every block count (measurement), bit 8 = QKV at <= 4 blocks as ONE {lo, hi} region x 256 rows per workgroup (fat waves
 *        of 2 x 2 tiles; default on), bit 9 = at 5-8 blocks too (measurement), bit 10 = the slab launches at <= 4 blocks in that one-region form (measured slower: opt-in),
 *        bit 11 = the paired gate/up fat launch stages its operands through registers (buffer_load -> VGPR -> ds_write) instead of LDS-DMA (measured slower: opt-in),
 *        bit 12 = QKV at <= 4 blocks over at most 128 regions (the fuller GQA image) as one region x 128 rows per workgroup (twice the workgroups of bit 8's form; default on),
 *        bit 13 = the fat slab launches (o_proj / down) with 2 / 4 / 8 K splits map ONE K split to an XCD (x per L2 = 1 / splits of it; same tiles, other workgroup ids; default on for grids of whole multiples of 256 workgroups, bit 14 = on any grid: measurement);
 *        bit-identical results; read at launch / capture.
 * key 7: idle-window weight prefetch of the 64-row step, KiB per workgroup of the next GEMM (0 = off, <= 128): the row kernels and
 *        the attention combine carry extra workgroups that pull the first k-tiles of the next GEMM into L2 (bit-identical
 *        results); key 8: start delay of those workgroups in s_sleep(32) rounds; key 9: KiB per down_proj workgroup pulled in from
 *        the tail of the gate/up launch (<= 64).  key 10: form of the tree-attention kernel, 0 = K/V tiles straight into the
 *        registers of both token-block waves, 1 = staged once per workgroup through LDS (LDS-DMA ring); bit-identical results.
 *        Keys 7-10 are read when a step graph is captured (la_llama_step captures again after a change).
 * key 11: the single-sequence step captured n (1..8) times into one graph (measurement of the per-launch cost: none found).
 * key 16: 1 = the gathered multi-block MoE step launches every expert's GEMMs separately and accumulates / normalises in two row
 *         kernels (default 0: one launch per stage, one fused row kernel); 2 = an expert's last single block keeps the padded
 *         two-block pass (default: the one-block body); 4 = the expert plan and the gather as two launches (round-3 form; default: one launch).
 * key 12: multi-block slab GEMMs with 2 K splits over 4 token groups at >= 5 blocks (measured slower; read at graph capture).
 * key 13: depth probe (also reachable through la_debug_set of the product header).  key 14: 1 = separate build-inputs / embed /
 *         argmax / accept / publish kernels instead of the fused step head / tail.  key 15: bit 0 = gate/up as 4 waves x 8 tile-sets.
 * key 17: tree attention of the single-sequence step, 1 = ONE launch (la_attn1.hip, default), 0 = key splits + combine.
 * key 18: variants of the single-launch attention (measurement): bit 0 = no start rotation, bits 1-2 = forced slice count.
 * key 19: residual + RMSNorm of the single-sequence step, 1 = four workgroups per row with a granule exchange (k_row_norm4:
 *         measured neutral, 5.30 vs 5.37 us per launch), 0 = one workgroup per row (k_row_norm, default).
 * key 20: multi-block tree attention, 1 = the next tile's V in flight through a per-wave LDS ring (LDS-DMA; measured slower), 0 = V
 *         requested when its tile begins (default); bit-identical results.
 * key 21: multi-block tree attention of GQA models, 1 = the query heads of a kv head start their key-tile lists at different offsets
 *         (changes the order of the online-softmax updates; measured neutral), 0 = all start at the first tile (default).
 * key 22: K splits of the gathered experts' down GEMM (0 = the engine's choice).  key 23: 1 = the slab GEMMs of the single-sequence
 *         step publish their partial sums write-through (measured slower: 4.23 vs 3.68 ms per step), 0 = plain stores (default).
 * key 24: schedule of the wide multi-block GEMMs, 1 = buffer-addressed LDS-DMA pieces and (>= 3 token tiles per wave) a fragment read
 *         after every MFMA (default), 0 = 64-bit per-lane piece pointers and the six reads of a k-tile together; bit-identical. 
 * key 25: merged-expert launches of the gathered MoE step as TWO workgroups per CU (4 instead of 8 weight tiles in flight per wave,
 *         <= 128 VGPRs): bit 0 = gate/up (default on), bit 1 = down_proj; round 5: TWO adjacent weight regions per workgroup (half the x
 *         traffic and LDS reads per weight byte; another fp32 summation order), bit 2 = gate/up (planned images), bit 3 = down_proj — both on
 *         by default (key 25 = 13), they take precedence over bits 0 / 1.
 * key 35: the paired gate/up launch of the multi-block step at 5-8 blocks, 1 = weights streamed straight into MFMA operand registers
 *         (k_gemm_fatd: four waves = four {gate, up} row-block pairs x all token tiles, only x through LDS; default), 0 = k_gemm_fat
 *         (weights and x through the LDS ring); bit-identical.
 * key 36: the slab launches (o_proj / down) of the multi-block step at 5-8 blocks, 1 = the x fragments of a wave's own token tiles streamed straight
 *         into MFMA operand registers, only the weights through the LDS ring (k_gemm_fat, STG = 2; default), 0 = both through the ring; bit-identical. */
int          la_lab_set(int key, int value);
int          la_lab_get(int key);          /* current value of a knob (the library default unless la_lab_set changed it) */
/* key 0: device buffer int64[workgroups][waves][8] the GEMM kernels stamp with wall_clock64() at entry / end of the
 * streaming loop / exit / half of the loop, plus HW_ID in word 4 (NULL = off). */
int          la_lab_set_ptr(int key, void* d_ptr);
#ifdef __cplusplus
}
#endif
#endif
