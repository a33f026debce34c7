/*
 * hipcomp/cascaded_select.h -- picking the options of the batched Cascaded codec from the data.
 *
 * AN API OF THIS LIBRARY'S OWN.  The reference tree has no selector for the batched interface
 * (its CHANGELOG mentions one for an earlier high-level API that is not in the tree): nothing here
 * replaces a reference function, and a caller that sticks to hipcomp/cascaded.h never sees it.
 * What it is for: hipcompBatchedCascadedOpts_t {num_RLEs, num_deltas, use_bp} decide the ratio of
 * the Cascaded codec on a column by an order of magnitude either way, and the
 * right values are a property of the data.
 *
 * How it decides: it does not guess from statistics, it MEASURES -- up to 64 partitions spread
 * evenly over the batch, the first 16 KiB of each, are compressed with every candidate option set
 * by the batched encoder itself into the temp buffer, and the set with the smallest total wins
 * (ties: the fewest layers).  Candidates: no compression at all; bit-packing alone; and
 * {num_RLEs 0..2} x {num_deltas 0..2} with bit-packing.  Cost: one tiny launch per candidate on
 * `stream` and ONE stream synchronisation (the result is returned to the host).
 */
#ifndef HIPCOMP_CASCADED_SELECT_H
#define HIPCOMP_CASCADED_SELECT_H

#include "hipcomp/cascaded.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of device temp space hipcompBatchedCascadedSelectOpts needs (independent of the batch). */
hipcompStatus_t hipcompBatchedCascadedSelectOptsGetTempSize(size_t* temp_bytes);

/*
 * device_uncompressed_ptrs / device_uncompressed_bytes: the batch as hipcompBatchedCascadedCompressAsync
 * takes it (device-accessible arrays of device pointers / sizes).  type: the element type of the columns.
 * device_temp_ptr / temp_bytes: at least ...GetTempSize bytes, 8-byte aligned.
 * opts_out (host): the chosen options, chunk_size 4096 and `type` filled in.  estimated_ratio (host,
 * may be NULL): uncompressed / compressed bytes of the sample under the chosen options.
 * Synchronises `stream`.  batch_size == 0: the default options, ratio 1.
 */
hipcompStatus_t hipcompBatchedCascadedSelectOpts(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipcompType_t type,
    void* device_temp_ptr,
    size_t temp_bytes,
    hipcompBatchedCascadedOpts_t* opts_out,
    double* estimated_ratio,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif
