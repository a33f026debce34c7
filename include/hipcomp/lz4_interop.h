/*
 * hipcomp/lz4_interop.h -- host-side helpers that make the batched LZ4 codec
 * interoperate with the LZ4 ecosystem on the CPU.
 *
 * hipcompBatchedLZ4CompressAsync produces raw LZ4 *blocks* (one per chunk), the
 * same format LZ4_decompress_safe() of liblz4 reads.  Files and streams in the
 * wild are LZ4 *frames* (magic 0x184D2204: header, length-prefixed blocks, end
 * mark).  These helpers wrap the blocks of a batch into one frame that the
 * `lz4` tool and LZ4F_decompress() read, and cut a frame made by anyone into
 * the block list that hipcompBatchedLZ4DecompressAsync takes.  The reference
 * lineage shipped this as examples (examples/lz4_cpu_compression.cu /
 * lz4_cpu_decompression.cu, removed from this tree: reference CHANGELOG.md:65-66);
 * examples/lz4_cpu_interop.c here shows both directions.
 *
 * Pure host code on host buffers (copy the compressed chunks and their sizes
 * back first): no GPU call, no stream.  Frames written: version 01, independent
 * blocks, no block / content checksums, content size present; frames read: any
 * frame whose blocks are independent (block checksums are skipped, a content
 * checksum is ignored, uncompressed blocks are reported as such).
 */
#ifndef HIPCOMP_LZ4_INTEROP_H
#define HIPCOMP_LZ4_INTEROP_H

#include "hipcomp/shared_types.h"

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes a frame of `num_chunks` blocks with `total_compressed_bytes` of block
 * data can take at most (header 15 + 4 per block + end mark 4). */
size_t hipcompLZ4FrameBound(size_t num_chunks, size_t total_compressed_bytes);

/*
 * Writes ONE frame holding the given LZ4 blocks in order.  Every chunk, and
 * every compressed block (incompressible data grows by n/255 + 16 bytes), must
 * be at most 4 MiB, the largest block size a frame can declare; the frame
 * declares the smallest of 64 KiB / 256 KiB / 1 MiB / 4 MiB that holds them all.
 * hipcompErrorInvalidValue: null pointer, chunk too large, frame_capacity too small.
 */
hipcompStatus_t hipcompLZ4FrameFromBlocks(
    const void* const* host_compressed_ptrs, const size_t* host_compressed_bytes,
    const size_t* host_uncompressed_bytes, size_t num_chunks, void* frame, size_t frame_capacity,
    size_t* frame_bytes);

/*
 * Parses the frame at `frame`: *num_blocks = number of data blocks,
 * *block_max_bytes = the block size the frame declares (the capacity to give
 * each chunk when decompressing).  With non-null arrays (of at least the
 * capacity passed in `blocks_capacity`) also fills, per block, the offset of
 * its data inside the frame, its stored size and whether it is stored
 * uncompressed (then copy it instead of decompressing it).
 * hipcompErrorCannotDecompress: not an LZ4 frame / truncated / bad header
 * checksum; hipcompErrorNotSupported: blocks depend on each other (linked
 * mode) or a dictionary is required; hipcompErrorInvalidValue: arrays too small.
 */
hipcompStatus_t hipcompLZ4FrameToBlocks(
    const void* frame, size_t frame_bytes, size_t* num_blocks, size_t* block_max_bytes,
    uint64_t* content_bytes /* nullable; 0 if the frame does not say */,
    size_t* block_offsets, size_t* block_bytes, int* block_is_raw, size_t blocks_capacity);

#ifdef __cplusplus
}
#endif

#endif
