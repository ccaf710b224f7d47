// l3_pipeline.hip -- the encode pipeline on gfx950 (wave64, LDS-staged tiles).
//
// Replaces the loops of llama3.Tokenizer.Encode (reference llama3/tokenizer.go:217-265) for a
// batch of documents held as one concatenated byte array + offsets:
//
//   docs_kernel    doc_off sanity; first document of every tile; clears the per-tile counters
//   tile_kernel    one workgroup per 3,840 bytes of text (+ halos) staged in LDS:
//                  UTF-8 validity + character class per byte    (state_machine.go:36,310-320)
//                  special-token literals                       (special.go:12,48-83)
//                  position-local start-of-pre-token predicate  (state_machine.go:82-284, l3_rules.h)
//                  whole-pre-token vocabulary probe             (bpe.go:27-33)
//                  one stream item per pre-token: its id, or a placeholder of a deferred pre-token
//   merge_kernel   one lane per short deferred pre-token (<= 64 bytes): greedy merge in an LDS slot (bpe.go:36-71);
//                  its last workgroups make the wave-per-entry pass over the slow list (mid_pass)
//                  and cut the long ones into window items (prep_entry)
//   slow_kernel    the remaining deferred pre-tokens (long, invalid UTF-8, or ending beyond the halo) as independent
//                  WINDOW items of 1,920 symbols: cut points where no vocabulary entry can span, every piece merged by
//                  the exact round-based parallel greedy merge, token ids into a sparse per-symbol image; pieces of
//                  32,768 symbols and more are listed for the grid-wide merge (l3_giant.hip, driven by the host)
//   block_totals_kernel   per-tile output bases inside blocks of 1,024 tiles and the totals of the blocks (the waves
//                  of the emit kernels add the block totals in front of their tile up themselves: no scan launch); the
//                  launch also carries the finish_window workgroups that compact the window images
//   emit_kernel    scatter to the final token array, per-document offsets, BOS/EOS (tokenizer.go:225-262);
//   emit_generic_kernel   the tiles the fast one flags
//   status_out_kernel     the status word to page-locked host memory, cleared for the next batch
//   decode_sum / decode_scan / decode_write_kernel   ids -> bytes in blocks of 4,096 ids (tokenizer.go:339-354)
//
// All cross-workgroup data flows through kernel boundaries on one stream; atomics are used only
// for counters / appends that later kernels read.
//
// The kernels live in l3_tile.hip, l3_merge.hip, l3_slow.hip, l3_emit.hip and l3_decode.hip (one translation unit per
// stage, each with the launch function of its stage); this file strings the stages of one batch together.
#include "l3_kernels.h"

namespace l3 {

#define L3_EV(i)                                                          \
    do {                                                                  \
        if (ev && ((ev_mask >> (i)) & 1u)) {                              \
            hipError_t e_ = hipEventRecord(ev[i], stream);                \
            if (e_ != hipSuccess) return e_;                              \
        }                                                                 \
    } while (0)

#define L3_STAGE(call)                       \
    do {                                     \
        hipError_t e_ = (call);              \
        if (e_ != hipSuccess) return e_;     \
    } while (0)

hipError_t launch_encode(const Tables& T, const Workspace& W, const EncodeArgs& A, hipStream_t stream, hipEvent_t* ev,
                         bool emit_only, uint32_t ev_mask) {
    const uint64_t n_tiles = num_tiles(A.n_bytes);
    if (n_tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (!emit_only) {
        L3_EV(kEvStart);
        L3_STAGE(launch_stage_docs(W, A, stream));
        L3_EV(kEvDocs);
        L3_STAGE(launch_stage_tile(T, W, A, stream));
        L3_EV(kEvTile);
        // slow_kernel's workgroups take entries from a queue: no more of them than the GPU holds at once (an empty list
        // then costs a short launch); the wave-per-entry pass inside merge_kernel's launch gets the same number
        uint64_t sgrid = W.slow_cap < L3_SLOW_GRID ? W.slow_cap : L3_SLOW_GRID;
        if (W.slow_grid && W.slow_grid < sgrid) sgrid = W.slow_grid;
        if (sgrid == 0) sgrid = 1;
        L3_STAGE(launch_stage_merge(T, W, A, (uint32_t)sgrid, stream));
        L3_EV(kEvMerge);
        L3_STAGE(launch_stage_slow(T, W, A, (uint32_t)sgrid, 0, stream));
        L3_EV(kEvSlow);
        L3_STAGE(launch_stage_totals(W, A, stream));
        L3_EV(kEvScan);
        L3_EV(kEvHost);
    }
    L3_STAGE(launch_stage_emit(W, A, stream));
    if (emit_only) {   // the re-run after the caller enlarged its buffer: the flags of the first run are gone
        L3_STAGE(launch_emit_generic(W, A, stream));
        L3_STAGE(launch_copy_runs(W, A, -1, -1, stream));
    }
    L3_EV(kEvEmit);
    return hipSuccess;
}

}  // namespace l3
