/* C entry point of the many-site host workflow in libparagraph_host.so (paragraph_amd/host):
 * what the reference's `grmpy` binary does for one graph list and one manifest
 * (src/c++/main/grmpy.cpp:60-260 -> grmpy::Workflow, src/c++/lib/grmpy/Workflow.cpp:71-199), as one call that a
 * driver like src/python/bin/multigrmpy.py can make instead of spawning `grmpy`.
 * Reads BAM files itself; realignment and counting run on the MI355X through libparagraph_amd.so.
 */
#ifndef PARAGRAPH_WORKFLOW_H
#define PARAGRAPH_WORKFLOW_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Genotypes every graph against every sample of the manifest and writes a JSON array with one genotype document per
 * graph (the document of GraphGenotyper::getGenotypes, in the order of graph_paths) to output_path.
 *
 *   reference_fasta        FASTA of the BAMs (with .fai, or it is scanned)
 *   manifest               grmpy manifest: id, path, depth + read length | idxdepth, [sex, depth sd, index_path]
 *   graph_paths/n_graphs   graph descriptions (JSON, share/schema/graph_schema.json)
 *   genotyping_parameters  grmpy -G document, or NULL / "" for the defaults
 *   options_json           NULL / "" or an object with any of: "threads", "lanes", "sites_per_batch", "max_reads",
 *                          "bad_align_frac", "path_sequence_matching", "kmer_sequence_matching", "klib_sequence_matching",
 *                          "exact_match_shortcut" (gssw-only cascade: reads whose record one exact full-length match forces skip
 *                          their fills, pg_batch_retire_exact_matches; same documents),
 *                          "bad_align_uniq_kmer_len", "packed_reads", "devices" (array of HIP device ordinals the
 *                          lanes are spread over, lane l on devices[l % n]; default: the PG_DEVICES environment
 *                          variable -- "0,1,2,3" or "all" -- else device 0; sites are independent, the devices
 *                          exchange nothing)
 *                          (grmpy's option names and defaults, grmpy/Parameters.hh:30-74; "sites_per_batch" = (graph,
 *                          sample) pairs per device batch, default 192 for every cascade; "lanes" = batches in flight, default about 1.5 per
 *                          host thread, at most 32 per device and at least one per device -- it grows with the device
 *                          list; a lane sleeps while its batch is on the device)
 *   error/error_cap        receives the message when the call fails (may be NULL)
 *
 * Returns 0 on success, 1 on failure (nothing usable is written then).
 */
int pgw_genotype_graphs(
    const char* reference_fasta, const char* manifest, const char* const* graph_paths, size_t n_graphs,
    const char* genotyping_parameters, const char* options_json, const char* output_path, char* error, size_t error_cap);

#ifdef __cplusplus
}
#endif
#endif
