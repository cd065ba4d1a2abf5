/*
 * pbgpu.h -- C-ABI of libpbgpu.so: the MI355X (gfx950) implementation of Platanus_B's k-mer counting and
 * de Bruijn graph hot path.
 *
 * The reference (rkajitani/Platanus_B v1.3.2) has no FFI/plugin interface: `Assemble` calls the C++ templates
 * `Counter<KMER>` and `BruijnGraph<KMER>` directly.  Every entry point below therefore names
 * the reference call it replaces (file:line relative to the reference tree).  A host that wants to be a
 * drop-in for `platanus_b assemble` keeps assemble.cpp's control flow and swaps those calls for these.
 *
 * Conventions
 *   - every function returns 0 on success or a negative PB_E* code; pb_last_error(ctx) gives the text;
 *   - no exceptions cross the boundary; plain pointers and sizes only;
 *   - the library owns all device memory, the caller owns all host memory (buffers are caller-allocated,
 *     sizes are obtained from the preceding call);
 *   - calls on one context are not re-entrant (the reference's only concurrency contract is "-t OpenMP threads
 *     inside one call", assemble.cpp:152);
 *   - one context = one GPU.  Multi-GPU = one context per process/rank; the exchange step hands device
 *     pointers to the caller's collective layer (RCCL through torch.distributed, or rccl directly).
 *
 * Key encoding (reference kmer.h:84-310, binstr.h:292-721): 2 bits/base A=0 C=1 G=2 T=3; a k-mer is a 2k-bit
 * integer with its 5' base in the most significant position, stored as W = ceil(k/32) little-endian 64-bit
 * words (word 0 = the 32 bases at the 3' end).  Keys handed over the boundary are canonical
 * (min(forward, reverse complement), counter.h:429) unless stated otherwise.
 */
#ifndef PBGPU_H
#define PBGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PB_OK            0
#define PB_EINVAL       -1   /* bad argument / call order                                   */
#define PB_EHIP         -2   /* HIP runtime error (text in pb_last_error)                   */
#define PB_ENOMEM       -3   /* device or host allocation failed                            */
#define PB_ERANGE       -4   /* k out of range (1..160), read too long (>= 500000, common.cpp:28) */
#define PB_ENODEVICE    -5   /* no gfx950 device visible                                    */
#define PB_EPEER        -6   /* N ranks: another rank has failed or died; the group was stopped (pb_dist_*) */

#define PB_MAX_K        160  /* KmerN<Binstr159>, assemble.cpp:182                          */
#define PB_OCC_BINS     65535 /* occurrenceDistribution.resize(UINT16_MAX), counter.h:335   */
#define PB_COUNT_MAX    65534 /* counts saturate at UINT16_MAX-1, counter.h:468-470         */

typedef struct pb_ctx pb_ctx;

/* ---- context ------------------------------------------------------------------------------------------- */
int         pb_create(int device_id, pb_ctx **out);
void        pb_destroy(pb_ctx *ctx);
const char *pb_last_error(const pb_ctx *ctx);          /* ctx may be NULL: error of the last failed pb_create */
/* Choice between equivalent implementations of a call (tests, A/B timing).  No option changes a result; the library reads no
 * environment variable for this.  Not in the reference (its only knobs are -t and -m).
 *   "count_path"      0 = by input size (default), 1 = direct pass (one lane per window, global table), 2 = super-k-mer pass (k <= 96:
 *                     one-word keys kernels_superk.inc, two-word keys kernels_superk2.inc, three-word keys kernels_superk3.inc; larger k
 *                     takes the direct pass anyway)
 *   "sk_bucket_bits"  0 = by input size, 2..20 = bucket bits of the super-k-mer pass;  "sk_l1_bits" 0 = by the bucket bits, 1..9
 *   "dist_design"     pb_dist_count: 0 = run records exchanged before the counting where that applies, 1 = always count + merge
 *   "merge_table" / "merge_16b"  1 = owner-side merge through the global table / with 16-byte records
 *   "chain_rank_full" 1 = pointer jumping over all chain states;  "grid_cap" n = cap on the grid of the direct counting kernel
 *   "verbose"         1 = one line per super-k-mer pass attempt on stderr (bucket bits, records, flagged buckets, spilled records)
 *   "sk_bucket_form"  2 (default) / 1 = which form of the bucket kernel counts a final bucket (k <= 32 and two-word keys have both)
 *   "sk_bucket_blocks" n = grid of the bucket kernel;  "graph" 1 (default) = the k <= 32 pass replayed as a captured HIP graph
 *   "hist_zeroed"     1 = the caller's histogram array is zero on entry (only the non-zero head is written)
 *   "sk_out_div"      n = (key, count) arrays of the k <= 32 pass sized for windows / n keys (large inputs: 5 by themselves; a pass
 *                     that finds more is repeated once with the number it counted) */
int         pb_check_option(const char *name, long value);   /* PB_OK if pb_set_option would take it (no context, no GPU: launchers check before they fork) */
int         pb_set_option(pb_ctx *ctx, const char *name, long value);
/* name, CU count, HBM bytes of the context's device (for reports; not in the reference) */
int         pb_device_info(const pb_ctx *ctx, char *name, size_t name_cap, int *n_cu, uint64_t *hbm_bytes);

/* ---- reads: replaces the per-thread temp files readFP[] (assemble.cpp:790-986, common.h:426-448) ------- */
/* Append reads given as one byte per base.  `ascii` != 0: bytes are FASTA/FASTQ letters and are converted
 * on the device with the reference's Char2Bin table (common.h:256); otherwise bytes are codes 0..3, 4 = N.
 * offsets[n_reads+1] are base offsets into `bases`.  Host buffers, copied before return.  On the device the
 * reads are packed to 2 bits/base plus a 1-bit unknown-base mask (kernel K1).  Append order = the order of
 * the reference's temp file 0 at -t 1 (file 1's reads, then file 2's, ...). */
int pb_reads_append(pb_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int ascii);
/* SURVEY 8(f) row 3 -- the read ingest on the device (replaces the host-side line parser in front of pb_reads_append; reference:
 * assemble.cpp:790-986, the FASTA / FASTQ line rules of Assemble::readFastaSingle... / readFastqSingle...).  `bytes` is a piece of an
 * uncompressed read file that starts at a record-mark line ('@' / '>' at a line start) -- or the file's first piece (first_piece != 0:
 * its first line is the format line checkFileFormat has looked at) --, smaller than 4 GiB; last_piece != 0: the piece ends the file (the
 * reference's unconditional trailing flush).  Newline positions, line classes, the sequence lines of every record and the read offsets are
 * found by map / prefix-scan kernels; the reads are then appended exactly as pb_reads_append(ascii = 1) appends the host parser's.
 * PB_ERANGE (nothing appended) if a read has 500000 bases or more.  pb_reads_starts: stream position of every read + final sentinel. */
int pb_reads_append_raw(pb_ctx *ctx, const uint8_t *bytes, uint64_t n_bytes, int is_fastq, int first_piece, int last_piece, uint64_t *n_reads, uint64_t *n_bases);
int pb_reads_starts(const pb_ctx *ctx, uint64_t *starts, uint64_t cap);
/* optional: announce how much pb_reads_append calls will bring in total, so that the packed stream is allocated once */
int pb_reads_reserve(pb_ctx *ctx, uint64_t n_bases, uint64_t n_reads);
int pb_reads_clear(pb_ctx *ctx);
int pb_reads_info(const pb_ctx *ctx, uint64_t *n_reads, uint64_t *n_bases);
/* lengthDistribution (counter.h:328-333): hist[len] for len < hist_n; *sum_len, *n_reads for the mean */
int pb_reads_length_hist(pb_ctx *ctx, uint64_t *hist, uint64_t hist_n);

/* ---- Counter::makeKmerReadDistributionMT (counter.h:275-383): K2+K3+K4 -------------------------------- */
/* Counts every N-free k-mer window of every resident read into a device occurrence table (canonical keys,
 * counts saturating at 65534), then builds occurrenceDistribution.  Outputs (any may be NULL):
 *   occ_hist[PB_OCC_BINS]  bin i = number of distinct k-mers seen i times (bin 0 unused), counter.h:482-507
 *   max_occ                highest non-empty bin, counter.h:371-376
 *   n_instances            windows counted;  n_distinct  distinct canonical k-mers                       */
int pb_count(pb_ctx *ctx, unsigned k, uint64_t *occ_hist, uint64_t *max_occ, uint64_t *n_instances, uint64_t *n_distinct);

/* ---- Counter::sortedKeyFromKmerFile + loadKmer (counter.h:916-951, 599-640): K4 compaction + K5 sort --- */
/* Keeps the k-mers with count >= min_occ, sorted ascending by the 2k-bit key value; *n_keys = how many. */
int pb_retain(pb_ctx *ctx, unsigned min_occ, uint64_t *n_keys);
/* Copies the retained set out: keys[n_keys * W] (word-major per key: key i = keys[i*W .. i*W+W-1]), counts[n_keys]. */
int pb_fetch_retained(pb_ctx *ctx, uint64_t *keys, uint16_t *counts);
/* All distinct (key,count) of the last pb_count, unsorted (for exchange/merge and for tests). */
int pb_fetch_all(pb_ctx *ctx, uint64_t *keys, uint32_t *counts, uint64_t cap, uint64_t *n);

/* ---- X1: multi-GPU merge of per-GPU occurrence tables (no reference counterpart: OpenMP threads share one
 * table behind 1024 stripe locks, counter.h:458-476).  Owner exchange: a key belongs to the rank
 * owner = ((key.word0 & 1023) * n_parts) >> 10, i.e. a contiguous range of the reference's 1024 stripes.
 *   pb_export_begin  buckets every (key,count) of the local table by owner on the device; part_counts[n_parts]
 *   pb_export_copy   copies the bucketed records (owner 0 first) into CALLER device buffers
 *                    dev_keys u64[total*W], dev_counts u32[total] -- e.g. torch tensors that then go through
 *                    torch.distributed.all_to_all_single (RCCL over xGMI)
 *   pb_merge_begin/add/end  owner side: fresh table, add received records (device pointers), then the merged
 *                    table becomes the context's table (pb_retain etc. work on it) and its histogram is returned;
 *                    the caller all-reduces the histograms (sum) to get the global occurrenceDistribution. */
int pb_export_begin(pb_ctx *ctx, int n_parts, uint64_t *part_counts);
int pb_export_copy(pb_ctx *ctx, void *dev_keys, void *dev_counts);
/* The same export with fewer bytes for the wire: inside an owner's segment the records with count 1 (most of a sequencing-depth
 * table: k-mers with a sequencing error, counter.h:464 "first sight = 1") come first and travel as bare keys, the others follow
 * with a 16-bit count (clamped to 65535; the sum saturates at 65534 at read-out, counter.h:468-470).
 *   pb_export_split_begin  single_counts[n_parts], multi_counts[n_parts]
 *   pb_export_split_copy   dev_keys u64[(singles + multis) * W] owner by owner (singles, then multis);
 *                          dev_counts16 u16[multis]: the multis of owner 0, of owner 1, ... */
int pb_export_split_begin(pb_ctx *ctx, int n_parts, uint64_t *single_counts, uint64_t *multi_counts);
int pb_export_split_copy(pb_ctx *ctx, void *dev_keys, void *dev_counts16);
int pb_merge_begin(pb_ctx *ctx, unsigned k, uint64_t expected_records);
int pb_merge_add(pb_ctx *ctx, const void *dev_keys, const void *dev_counts, uint64_t n);
int pb_merge_end(pb_ctx *ctx, uint64_t *occ_hist, uint64_t *max_occ, uint64_t *n_distinct);
/* pb_merge_begin + one pb_merge_add + pb_merge_end in one call; for k <= 32 and >= 1 M records the records are merged through
 * the counting pass's partition + LDS tables instead of a global table (same result, no random device-scope atomics). */
int pb_merge_records(pb_ctx *ctx, unsigned k, const void *dev_keys, const void *dev_counts, uint64_t n, uint64_t *occ_hist,
                     uint64_t *max_occ, uint64_t *n_distinct);

/* ---- X1, second design: records exchanged BEFORE they are counted (k <= 32).  The minimizer bucket of a k-mer is a function of the
 * k-mer, so "owner = top bits of the bucket" partitions the key space like the reference's stripes do.
 *   pb_sk_bucket_bits       the bucket bits the library would use for `windows` k-mer instances on one GPU (every rank passes the same
 *                           local_bits, e.g. pb_sk_bucket_bits(max over ranks of the windows a rank will receive))
 *   pb_sk_owner_records     this rank's reads cut into 16-byte run records grouped by owner; counts[o] = records for owner o
 *                           (n_owners a power of two >= 2, log2(n_owners) + local_bits <= 20)
 *   pb_sk_owner_records_copy  the records, owner 0 first, into a caller device buffer (e.g. a torch tensor that then goes through
 *                           torch.distributed.all_to_all_single)
 *   pb_sk_count_records     counts the records a rank received (any order): outputs as pb_count's for this rank's share of the keys;
 *                           afterwards pb_retain / pb_fetch_all / pb_export_* work on it */
unsigned pb_sk_bucket_bits(uint64_t windows);
int pb_sk_owner_records(pb_ctx *ctx, unsigned k, unsigned n_owners, unsigned local_bits, uint64_t *counts, uint64_t *n_windows);
int pb_sk_owner_records_copy(pb_ctx *ctx, void *dev_records);
int pb_sk_count_records(pb_ctx *ctx, unsigned k, const void *dev_records, uint64_t n_records, unsigned local_bits, uint64_t *occ_hist, uint64_t *max_occ,
                        uint64_t *n_distinct);

/* ---- E2: N ranks of one node on one run (configs 4 and 5: reads range-partitioned over the GPUs, one merged table for the graph
 * stage and for every later round, assemble.cpp:343-354,449-478).  The launcher calls pb_dist_shared_create BEFORE it forks the ranks
 * (no HIP call inside); every rank then makes its own context and attaches.  All pb_dist_* calls below are collective: every rank
 * of the group must make them in the same order.
 *   pb_dist_attach         use_rccl != 0: records travel device to device over RCCL/xGMI (one GPU per rank; librccl.so.1 is loaded
 *                          on demand); 0: through shared host memory (ranks that share a GPU)
 *   pb_dist_merge_counts   after pb_count on this rank's reads: owner exchange + merge; the context's table is then this rank's owner
 *                          shard, the outputs describe the GLOBAL distribution (= pb_count's outputs of a single process on all reads)
 *   pb_dist_gather_retained  after pb_retain on every shard: the union of all shards, sorted, becomes every rank's retained set
 *   pb_dist_count_considering_contigs  pb_count_considering_contigs over all ranks' surviving reads (the contig table is replicated)
 *   pb_dist_allreduce_u64  element-wise sum of a host array (read-length histogram, kept-read counts) */
int pb_dist_shared_create(int world, void **shared);
/* Launcher side of an N-rank run (no HIP call in any of these; no reference counterpart: the reference is one process).
 *   pb_dist_launch         forks world - 1 ranks BEFORE the process has touched the GPU and returns in every process with its rank;
 *                          children die with the launcher (PR_SET_PDEATHSIG), the launcher gets a thread that reaps them and stops
 *                          the group (every barrier returns PB_EPEER) as soon as one ends abnormally
 *   pb_dist_mark_failed    async-signal-safe: stops the group (a rank's crash handler calls it before re-raising the signal)
 *   pb_dist_report_failure a rank that cannot go on: its exit status and the one line for stderr; stops the group
 *   pb_dist_finish         launcher, after its own run: waits for the ranks (kills what does not leave a failed group within 10 s) and
 *                          returns the run's exit status and its ONE error message (own failure, else the first rank's own reason,
 *                          else "rank r was killed by signal s")
 *   pb_dist_shared_barrier the group's barrier without a context (tests) */
int  pb_dist_launch(void *shared, int *rank);
void pb_dist_mark_failed(void *shared);
int  pb_dist_failed(void *shared);
int  pb_dist_report_failure(void *shared, int rank, int code, const char *msg, int secondary);
int  pb_dist_finish(void *shared, int own_code, const char *own_msg, int own_secondary, int *exit_code, char *msg, size_t msg_cap);
int  pb_dist_shared_barrier(void *shared, int rank);
int pb_dist_attach(pb_ctx *ctx, void *shared, int rank, int world, int use_rccl);
int pb_dist_detach(pb_ctx *ctx);
int pb_dist_info(const pb_ctx *ctx, int *rank, int *world, int *rccl);
int pb_dist_barrier(pb_ctx *ctx);
int pb_dist_allreduce_u64(pb_ctx *ctx, uint64_t *buf, uint64_t n);
/* the RCCL calls behind the functions below, with this rank as its own peer (n_bytes sent and received in one group, compared): what a
 * one-GPU box can check of the RCCL data path */
int pb_dist_rccl_selftest(pb_ctx *ctx, uint64_t n_bytes);
int pb_dist_merge_counts(pb_ctx *ctx, uint64_t *occ_hist, uint64_t *max_occ, uint64_t *n_distinct);
/* pb_count over all ranks' reads in one call: for k <= 32 and a power-of-two number of ranks the run records are exchanged BEFORE they are
 * counted (pb_sk_owner_records / pb_sk_count_records: no export, no merge); otherwise pb_count + pb_dist_merge_counts */
int pb_dist_count(pb_ctx *ctx, unsigned k, uint64_t *occ_hist, uint64_t *max_occ, uint64_t *n_instances, uint64_t *n_distinct);
int pb_dist_count_considering_contigs(pb_ctx *ctx, unsigned k, uint64_t *occ_hist, uint64_t *max_occ, uint64_t *n_distinct);
int pb_dist_gather_retained(pb_ctx *ctx, uint64_t *n_total);

/* ---- graph stage: device part of BruijnGraph::makeInitialBruijnGraph (graph.h:289-579) ------------------------ */
/* On the retained, sorted set of the last pb_retain (n keys).  For retained k-mer i in its canonical orientation:
 *   adj[i]      = (left flags << 4) | right flags; bit b of a nibble is set iff the predecessor / successor with
 *                 base b is retained (the 8 counter.findValue probes of graph.h:352-380)
 *   nbr[2i]     right side, nbr[2i+1] left side: when exactly one flag is set on that side,
 *                 (index of that neighbour in the sorted set << 1) | flip, flip = 1 if the neighbour k-mer as reached is
 *                 the reverse complement of its canonical key; 0xFFFFFFFF otherwise.
 * With these two arrays the host replays the reference's DFS (visiting order, strands, cut points of cycles) in O(n)
 * array steps instead of hash probes; see platanus_b_amd/host/device_graph.hpp. */
int pb_graph_adjacency(pb_ctx *ctx, uint8_t *adj, uint32_t *nbr);   /* adj = nbr = NULL: keep the arrays on the device only */

/* K7 chain compaction (the unitig walk of graph.h:406-542 as node-parallel kernels: pointer-jumping list ranking over
 * the adjacency arrays, per-chain coverage sum, sequence gather).  Call after pb_graph_adjacency.
 *   chain_of[n]  chain id of k-mer i | 0xFFFFFFFF junction k-mer | 0xFFFFFFFE k-mer on a cycle / hairpin / self loop
 *                (the host walks those k-mer by k-mer with the reference's "visited" rule)
 *   orient[n]    1 if k-mer i is traversed as its reverse complement when the chain is read from end A to end B
 *   per chain c: len[c] k-mers, sum[c] = sum of their counts, end_a[c] / end_b[c] = (k-mer index << 1) | traversed-as-revcomp
 *                of the first / last k-mer, seq_woff[c] = first word of its packed sequence in seq[] (A -> B string of
 *                len+k-1 bases in Straight::seq layout: position p counts from the B end, 32 positions per word)
 * Chain numbering is arbitrary (it does not reach any output: the host replays the reference's DFS order). */
int pb_graph_chains(pb_ctx *ctx, uint64_t *n_chains, uint64_t *seq_words);
int pb_graph_fetch_chains(pb_ctx *ctx, uint32_t *chain_of, uint8_t *orient, uint32_t *len, uint64_t *sum, uint32_t *end_a,
                          uint32_t *end_b, uint64_t *seq_woff, uint64_t *seq);

/* Compact form of the same hand-over (what `platanus_b assemble` uses): the reference's DFS (graph.h:330-579) only ever
 * lands on junction k-mers, k-mers on cycles / hairpins and the two end k-mers of a simple chain, so only those "nodes" and
 * the per-chain records cross PCIe; the retained keys and the per-k-mer arrays (8W + 15 bytes per k-mer) stay on the device.
 * Call after pb_graph_chains.  Per node s (arbitrary order):
 *   idx[s]          index of the k-mer in the retained set
 *   meta[s]         adj | orient << 8 | count << 16        (adj, orient as above)
 *   chain[s]        chain id | 0xFFFFFFFF junction | 0xFFFFFFFE cycle / hairpin
 *   nbr[2s], [2s+1] as pb_graph_adjacency
 *   nb8[8s + b]     junctions only: (index << 1 | flip) of the successor with base b (b < 4) / predecessor with base b - 4 of
 *                   the canonical key, 0xFFFFFFFF if absent (the eight findValue probes of graph.h:352-380 with their results)
 *   keys[s * W ..]  canonical key
 * Per chain, additionally to pb_graph_fetch_chains: root[c] = (smallest member index << 1) | orient of that member, i.e.
 * where the reference's ascending-key root loop first meets the chain. */
int pb_graph_nodes(pb_ctx *ctx, uint64_t *n_nodes);
int pb_graph_fetch_nodes(pb_ctx *ctx, uint32_t *idx, uint32_t *meta, uint32_t *chain, uint32_t *nbr, uint32_t *nb8, uint64_t *keys);
int pb_graph_fetch_chain_records(pb_ctx *ctx, uint32_t *len, uint64_t *sum, uint32_t *end_a, uint32_t *end_b, uint32_t *root,
                                 uint64_t *seq_woff, uint64_t *seq);

/* ---- graph cleaning: the order-independent decisions as node-parallel kernels -------------------------------- */
/* cutBranch (graph.h:829-926) and, when coverage_cut != 0, deleteErroneousStraightNode (graph.h:2032-2082) over the live nodes of the
 * caller's graph: straight i has head / tail k-mer (strand-specific, W words each, word 0 least significant), length (k-mers), coverage
 * and out flags (left nibble << 4 | right nibble, one-hot); tail_entry_* are the entries of rightStraightTable as the caller holds them
 * (tail key -> straight index); junction j has key, coverage and out flags.  decision[i]: bit 0 = cutBranch
 * deletes straight i (a tip of length <= k hanging on a junction whose other neighbours on that side reach more than
 * coverage / branch_threshold), bit 1 = the erroneous-node rule does (coverage < coverage_cut and length < length_cut, not cut as a tip).
 * The caller applies the deletions in its container order (findStraightHead / findStraightTail / junctionTable look-ups of graph.h:194-219
 * are done on the device through one table of tagged keys).  Order-independent for branch_threshold < 1 only. */
int pb_nodes_clean_decide(pb_ctx *ctx, unsigned k, uint64_t n_straight, const uint64_t *head_keys, const uint64_t *tail_keys, const uint64_t *length,
                          const uint16_t *coverage, const uint8_t *out, uint64_t n_tail_entries, const uint64_t *tail_entry_keys, const uint32_t *tail_entry_ids,
                          uint64_t n_junction, const uint64_t *junction_keys, const uint16_t *jcoverage, const uint8_t *jout, double branch_threshold,
                          uint64_t length_cut, uint64_t coverage_cut, uint8_t *decision);
/* crushBubble's pairwiseAlignment (graph.h:712-746) for n_pairs pairs of base strings (codes 0..3): pair p = bases[offsets[2p] .. offsets[2p+1])
 * against bases[offsets[2p+1] .. offsets[2p+2]); dist[p] = unit-cost edit distance. */
int pb_edit_distances(pb_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_pairs, uint64_t *dist);

/* ---- the local de Bruijn graphs of gap closing, a batch of gaps per call ------------------------------------------
 * Replaces, for every gap of the batch at once, Counter<KMER>::countKmerForGapClose (counter.h:1015-1056: the FORWARD k-mer of every window
 * without an unknown base, counts saturating at 65534; called per gap at gapCloseDBG.h:91-120) and
 * BruijnGraph<KMER>::makeBruijnGraphForGapClose (graph.h:2116-2345: strand-specific junction / straight nodes over the k-mers with
 * value >= min_coverage, start k-mers in ascending key order: counter.h:168-175; called at gapCloseDBG.h:132-161).
 * Gap g owns the sequences seq_first[g] .. seq_first[g+1]-1; sequence s = codes[seq_off[s] .. seq_off[s+1]) (0..3, 4 = unknown).  The caller
 * passes the sequences that are counted: the reference skips a gap's first GAP_CANDIDATE_SEQ_1ST = 2 entries (counter.h:1023).
 * pb_gap_graphs_fetch: the nodes gap by gap, IN THE ORDER of the reference's setJunctionTable / insertStraight calls (inserted in that order
 * into tables with the reference's hashers they iterate as the reference's do: tests/golden/gapclose_*.txt are dumps of the reference's own
 * tables).  node_first[n_gaps+1]; per node kind (1 straight, 0 junction), coverage, out ((left flags << 4) | right flags), length in k-mers;
 * word_off[n_nodes+1] into words: a straight's sequence ((length + k + 31) / 32 words, base p from the 3' end at bits 2p: graph.h:67-71,
 * 2320-2338), a junction's k-mer (ceil(k / 32) words, least significant first). */
int pb_gap_graphs_build(pb_ctx *ctx, unsigned k, long min_coverage, uint64_t n_gaps, const uint64_t *seq_first, uint64_t n_seqs, const uint64_t *seq_off,
                        const uint8_t *codes, uint64_t *n_nodes, uint64_t *n_words);
int pb_gap_graphs_fetch(pb_ctx *ctx, uint64_t *node_first, uint8_t *kind, uint16_t *coverage, uint8_t *out, uint32_t *length, uint64_t *word_off, uint64_t *words);
/* mergeGraph + the graph over the merged table (gapCloseDBG.h:166-258, 424-425; BruijnGraph<KMER>::saveLargeKmerForGapClose, graph.h:2347-2370):
 * gap g owns the straight nodes node_first[g] .. node_first[g+1]-1 IN THE ORDER in which the reference's two saveLargeKmerForGapClose calls walk
 * them (the low graph's leftStraightTable, then the high graph's; nodes with coverage 65535 left out); node s has n_bases[s] bases (its length +
 * its own k - 1), coverage[s] and its sequence words at word_off[s].  Every k-mer of every node goes into the gap's table with the coverage of
 * the LAST node that holds it (setOccurrenceValueGap overwrites), then the nodes over that table as in pb_gap_graphs_build; pb_gap_graphs_fetch. */
int pb_gap_graphs_build_from_nodes(pb_ctx *ctx, unsigned k, long min_coverage, uint64_t n_gaps, const uint64_t *node_first, uint64_t n_in, const uint32_t *n_bases,
                                   const uint16_t *coverage, const uint64_t *word_off, const uint64_t *words, uint64_t *n_nodes, uint64_t *n_words);

/* ---- multi-k rounds (assemble.cpp:392-498) ------------------------------------------------------------------- */
/* saveEdgeKmer + pickupReadMatchedEdgeKmer (graph.h:1877-1939, counter.h:869-910): keep a read iff one of its N-free
 * k-mers (k = the finished graph's k) is among edge_keys[n_edge*W] (canonical).  Cumulative like the reference's readFP:
 * a dropped read stays dropped until pb_reads_restore (= allReadFP, assemble.cpp:164-171).  Dropped reads are masked
 * out of every later counting call. */
int pb_reads_filter(pb_ctx *ctx, unsigned k, const uint64_t *edge_keys, uint64_t n_edge, uint64_t *n_kept);
int pb_reads_restore(pb_ctx *ctx);
/* saveContig<NEXT> (graph.h:1655-1771): every next-k k-mer of contig string c (codes 0..3, offsets[n_contigs+1]) gets
 * value = max(value, values[c]); values[c] == 0 writes nothing.  The table stays resident for the next call. */
int pb_contig_table_build(pb_ctx *ctx, unsigned k, const uint8_t *bases, const uint64_t *offsets, const uint16_t *values,
                          uint64_t n_contigs, uint64_t *n_windows);
/* makeKmerReadDistributionConsideringPreviousGraph (counter.h:662-750): k-mers of the (filtered) reads that are absent
 * from the contig table are counted; result set = contig table entries (contig coverage) + those read k-mers.
 * Afterwards pb_retain / pb_fetch_* / pb_graph_adjacency work on that set. */
int pb_count_considering_contigs(pb_ctx *ctx, unsigned k, uint64_t *occ_hist, uint64_t *max_occ, uint64_t *n_distinct);

/* ---- CT8: loadKmer + outputOccurrenceTableBinary (counter.h:599-640,954-963; doubleHash.h:266-278), k <= 32 ------- */
/* Reproduces the slot every retained k-mer occupies in the reference's final table at -t 1: first-occurrence order into
 * the 1024 stripe tables of dh_size/1024 slots with the 32(+)-probe spill passes (counter.h:340-364,458-476), stripe-major
 * dump order, then insertion of the k-mers with count >= min_occ into a table of max(need, dh_size) slots.
 * dh_size = the value makeKmerReadDistributionMT returns for the run's -m (a power of two; host arithmetic, counter.h:300-309).
 * Outputs: *table_size (the header writes table_size-1), *n_records; pb_kmer_occ_fetch copies the records in ascending
 * slot order (exactly what writeTable() emits): slots[n], keys[n], counts[n]. */
int pb_kmer_occ_layout(pb_ctx *ctx, uint64_t dh_size, unsigned min_occ, uint64_t *table_size, uint64_t *n_records);
int pb_kmer_occ_fetch(pb_ctx *ctx, uint64_t *slots, uint64_t *keys, uint16_t *counts);

/* ---- SURVEY 8(f) "next" row 1: `kmer_divide`'s look-ups (ContigDivider::getOccurrenceArray, kmer_divide.cpp:150-197) ----- */
/* pb_occ_table_load: the (key,count) records of a PREFIX_kmer_occ.bin (read on the host) become the resident lookup table.
 * pb_occ_lookup: sequences given as codes 0..3 / 4 = N with offsets[n_seqs+1]; out[offsets[i] - offsets[0] + j] = occurrence of the
 * canonical k-mer starting at position j of sequence i, 0 if absent, if the window covers an N or if j > len_i - k. */
int pb_occ_table_load(pb_ctx *ctx, unsigned k, const uint64_t *keys, const uint16_t *counts, uint64_t n);
int pb_occ_lookup(pb_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_seqs, uint16_t *out);

/* ---- timing of the last call, measured with HIP events on the library's stream ------------------------- */
/* stage names: "pack", "count", "hist", "compact", "sort", "adjacency", ... ; returns PB_EINVAL if unknown */
int pb_last_kernel_ms(const pb_ctx *ctx, const char *stage, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* PBGPU_H */
