/*
 * capdec_tuning.h -- kernel-selection overrides of the TUNING build of the library (libcapdec_tune.so).
 *
 * The product library libcapdec.so is compiled without them: there the choices below are compile-time constants and
 * the library keeps no state between calls.  The tuning build is the same sources compiled with
 * -DCAPDEC_TUNING_HOOKS; it exports everything capdec.h declares plus these six process-wide switches, which the
 * tests use to run one input through two kernel variants (and compare them bit for bit) and the benchmark tools use to
 * time tile shapes.  They have no counterpart in the reference.  Not thread-safe; never link this build into a server.
 */
#ifndef CAPDEC_TUNING_H
#define CAPDEC_TUNING_H

#ifdef __cplusplus
extern "C" {
#endif

/* contraction tiles: 0 = automatic, 11 / 116 = 64x64 (BK 32 / 16), 22 / 216 = 128x128 (BK 32 / 16), 32 / 316 = 32x64 tile
 * with its k-loop split over the two wave pairs (BK 32 / 16) */
int capdec_gemm_force_config(int cfg);
/* 0 = automatic, 1 = operand tiles always staged through registers (never loaded straight into LDS; also keeps
 * batches of at most 32 rows of the vocabulary arg-max on the general kernel) */
int capdec_gemm_force_staging(int mode);
/* the same choice for the LSTM step kernel; 2 = batches of at most 32 samples keep the kernel that reads its MFMA
 * fragments straight from memory (E and Hd multiples of 32 normally take the one with whole-line loads) */
int capdec_lstm_force_staging(int mode);
/* non-zero: attention backward always takes the general two-pass kernels (sequences up to 64 positions normally take a
 * one-pass kernel that keeps a (batch, head) slice in LDS) */
int capdec_attention_bwd_force_two_pass(int on);
/* non-zero: the self-attention backward of a ragged batch keeps the row-by-row one-pass kernel where the matrix-core
 * form (head width 128, at most 32 positions) would be taken */
int capdec_attention_bwd_force_row_form(int on);
/* non-zero: long unmasked memories (more than 64 keys) stay on the one-wavefront-per-row forward kernel instead of the
 * chunked one */
int capdec_attention_force_row_kernel(int on);
/* non-zero: the causal self-attention of a ragged batch keeps the row-by-row kernel where the matrix-core form (head
 * width 128, at most 32 positions) would be taken */
int capdec_attention_ragged_force_row_form(int on);
/* non-zero: the KV-cached beam search keeps a step's decoder layers as separate launches (shapes that the one-launch
 * layers of the greedy decoder support normally take those) */
int capdec_beam_force_unfused(int on);

#ifdef __cplusplus
}
#endif
#endif /* CAPDEC_TUNING_H */
