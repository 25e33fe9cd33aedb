/*
 * c3hip.h -- C ABI of libc3hip.so, the MI355X (gfx950) native inference path for the two Clair3
 * networks.  This is the drop-in boundary for the *model call* of the reference's
 * CallVariantsFromCffi / CallVariantsFromCffiGPU step; everything else of the reference pipeline
 * (tensor extraction, VCF emitters, run_clair3.py) stays untouched.
 *
 * What each entry point replaces in the reference (paths relative to the Clair3 repo):
 *
 *   c3_device_count / c3_mem_info   nvidia-smi parsing in clair3/CallVariantsFromCffiGPU.py:13-43
 *                                   (get_gpu_memory / check_gpu_memory) -- not available on ROCm.
 *   c3_model_create                 model factory  clair3/CallVariantsFromCffi.py:223-243
 *                                   (Clair3_P / Clair3_F(add_indel_length, predict=True, input_channels)),
 *                                   clair3/model.py:58-128 and :282-368; device selection :215-221.
 *   c3_model_load                   _load_torch_checkpoint + strict load_state_dict,
 *                                   clair3/CallVariantsFromCffi.py:19-28 (the Python wrapper does the
 *                                   torch.load and hands the named float32 tensors over).
 *   c3_predict                      _torch_predict(model, device, X) -> numpy (B, 24|90) float32,
 *                                   clair3/CallVariantsFromCffi.py:48-52 (twin: clair3/CallVariants.py:83-87):
 *                                   H2D copy + Clair3_P.forward (model.py:130-161) or Clair3_F.forward
 *                                   (model.py:377-416) + D2H copy.
 *   c3_predict_submit / _wait       same, split so the caller's loop (CallVariantsFromCffi.py:302-353)
 *                                   can overlap batch i+1's transfer with batch i's kernels/decoding.
 *   c3_predict_device               the forward pass alone on tensors already resident in HBM
 *                                   (what bench.py times; also the hook for the RCCL gather of SURVEY 8e).
 *   c3_model_destroy                model going out of scope at process exit.
 *   c3_vcf_rows                     the per-row Python of the decoder: batch_output -> output_with -> output_from with its allele
 *                                   lookups (clair3/CallVariants.py:1069-1394, :676-1016, :117-201, :662-673) for rows that carry the
 *                                   decoder columns -- one pass of plain host code per batch.
 *   c3_device_pci_bus_id            where a GPU slot's worker belongs on the host (the reference leaves placement to the OS,
 *                                   clair3/CallVariantsFromCffiGPU.py:138-156).
 *
 * Conventions (mirroring libclair3's cffi surface, build.py:38-85, src/clair3_pileup.h:90-113):
 *   - plain C types only; the library owns all device memory; the caller owns host x / y buffers
 *     (C-contiguous; pageable is fine -- the library stages through its own pinned buffers);
 *   - every int-returning function returns 0 on success, non-zero on failure, and c3_last_error()
 *     then describes the failure (thread-local).  Nothing aborts the process;
 *   - one model per handle, one handle per OS process per GPU slot is the intended use
 *     (clair3/CallVariantsFromCffiGPU.py:163-199 launches one worker per slot); a handle is not
 *     thread-safe, different handles are independent.
 *   - the output row layout is exactly the reference's: [gt21 (21) | genotype (3) | indel_1 (33) |
 *     indel_2 (33)], columns label_shape_cum = 21,24,57,90 (shared/param_p.py:37-39), float32
 *     probabilities, consumed unchanged by batch_output (clair3/CallVariants.py:1069-1116).
 */
#ifndef C3HIP_H
#define C3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C3_KIND_PILEUP 0         /* Clair3_P, windows (B, 33, C)    int8 | int32 */
#define C3_KIND_FULL_ALIGNMENT 1 /* Clair3_F, windows (B, D, 33, C) int8, channels-last */

#define C3_DTYPE_I8 0
#define C3_DTYPE_I32 1
#define C3_DTYPE_F32 2
#define C3_DTYPE_I64 3

typedef struct c3_model c3_model;

/* one entry of a PyTorch state_dict */
typedef struct {
    const char *name;  /* e.g. "LSTM1.weight_ih_l0", "conv1.bn.running_var" */
    int32_t dtype;     /* C3_DTYPE_F32 for every parameter/buffer; *.num_batches_tracked (I64) is accepted and ignored */
    int32_t ndim;
    int64_t shape[4];
    const void *data;  /* host pointer, C-contiguous */
} c3_tensor_desc;

const char *c3_version(void);
/* thread-local description of the last failure in this thread ("" if none) */
const char *c3_last_error(void);

/* number of visible HIP devices (honours HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES); <0 on error */
int c3_device_count(void);
int c3_mem_info(int device, size_t *free_bytes, size_t *total_bytes);
/* PCI address of a visible device as sysfs spells it ("0000:c5:00.0"): /sys/bus/pci/devices/<address>/numa_node is the NUMA
 * node the host side of a rank -- staging copies, the forked decode workers -- belongs on (clair3_amd/dist.py pin_to_device_numa).
 * The reference starts one worker process per GPU slot and leaves placement to the OS (clair3/CallVariantsFromCffiGPU.py:138-156,
 * parallel -j); SURVEY 8e names NUMA placement as a limiter of the 1 -> 8 GPU scaling. */
int c3_device_pci_bus_id(int device, char *buf, int buf_bytes);

/* kind: C3_KIND_*; in_channels: 18 (pileup) / 8 or 9 (full alignment, 9 = dwell time);
 * add_indel_length: 0 -> (B,24) output, 1 -> (B,90).  Returns NULL on failure. */
c3_model *c3_model_create(int kind, int in_channels, int add_indel_length, int device);
/* window geometry; defaults are the ONT shapes: depth 89 (ignored for pileup), 33 positions */
int c3_model_set_geometry(c3_model *m, int depth, int positions);
/* strict: every expected key must be present with the expected shape, unknown keys are an error */
int c3_model_load(c3_model *m, const c3_tensor_desc *tensors, int n_tensors);
/* 24 or 90 */
int c3_model_output_size(const c3_model *m);
/* Decoder columns: with enable != 0 every output row of c3_predict / c3_predict_submit /
 * c3_predict_device / c3_predict_pileup_region grows from c3_model_output_size() to c3_model_row_size() =
 * output_size + C3_DECODE_COLS floats.  The caller of _torch_predict (clair3/CallVariantsFromCffi.py:48-52, :317)
 * does not know the reference base of a row, so the columns carry what clair3/CallVariants.py:510-659 + :722-749
 * derive from the row for EVERY base (same float32 products, same order as c3_outcome_maxima):
 *   [0..8]   max of the class lists homo_SNP, hetero_SNP, homo_Ins, homo_Del, hetero_ACGT_Ins, hetero_InsIns,
 *            hetero_ACGT_Del, hetero_DelDel, hetero_InsDel        [9..12] homo_Ref probability if the base is A, C, G, T
 *   [13..21] position of the first occurrence of each maximum     [22]    bit b set: base b takes the early exit
 *   [23..26] the class output_from settles on first (:722-751) if the base is A, C, G, T: 0 = homo_Ref (the overall maximum, or
 *            the early exit), else the first class of its if / elif chain (:753-978) whose list holds the overall maximum
 *   [27..30] 100 x QUAL of that first decision (quality_score_from, :375-381), an integer: QUAL = column / 100.0
 * (positions, classes, bits and 100 x QUAL stored as float values).  Rows stay valid input of the reference's batch_output: it slices
 * columns [0:21] [21:24] [24:57] [57:90] (CallVariants.py:1072-1080) and never looks further right. */
#define C3_DECODE_COLS 31
int c3_model_set_decode_columns(c3_model *m, int enable);
int c3_model_row_size(const c3_model *m);
/* bytes of one input window for dtype x_dtype (594 / 2376 / 23496 / 26433 for the ONT shapes) */
int64_t c3_model_window_bytes(const c3_model *m, int x_dtype);

/* y_host[batch][24|90 (c3_model_row_size)] = forward(x_host[batch][...]); synchronous.  A batch of two or more chunks (256
 * full-alignment / 4096 pileup windows) travels through slots 0..2 of the submit / wait ring below in growing pieces:
 * no c3_predict_submit of this handle may be pending on those slots. */
int c3_predict(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host);
/* Note on the arithmetic: the contractions form their fp32 products from two fp16 pieces per operand (fp16x3, DESIGN.md 1:
 * fp32-level parity).  Should a checkpoint ever drive an activation towards the fp16 range (|x| >= 16000), c3_predict /
 * c3_predict_wait notice (a flag raised by the kernels, or a non-finite row), print one line to stderr, switch the handle
 * to the fp32 matrix instructions for the rest of its life and run the batch again; c3_predict_device does not check
 * (c3_predict_device_checked does; c3_model_range_status reports). */
#define C3_HOST_SLOTS 4
/* asynchronous pair, slot in [0, C3_HOST_SLOTS): submit copies x into pinned staging and enqueues H2D + kernels + D2H;
 * wait blocks until y_host of that slot is complete. x_host may be reused as soon as submit returns.  Batches in different slots
 * may run side by side on the device (the handle keeps up to three lanes -- workspace + streams -- for batches that do not fill the
 * chip by themselves; C3HIP_RING_LANES): rows never depend on the slot, the lane or the batch a window travels in. */
int c3_predict_submit(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host, int slot);
int c3_predict_wait(c3_model *m, int slot);
/* The same ring with the rows LEFT ON THE DEVICE: y_dev is a device pointer on the model's device (batch x c3_model_row_size()
 * floats) that the forward pass writes directly; nothing but the range flag crosses PCIe on the way out.  For a rank of a
 * sharded job whose rows go to c3_gather_rows, not to its own host -- the reference's per-GPU workers each write their rows to
 * disk (clair3/CallVariantsFromCffiGPU.py:138-199); here they meet on rank 0 over xGMI and cross PCIe once, there.
 * c3_predict_wait(slot) runs the range guard (flag + a device-side scan for non-finite rows) and re-runs on fp32 if needed. */
int c3_predict_submit_dev(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_dev, int slot);
/* There is deliberately NO entry that page-locks caller memory.  Rounds 3-5 exported c3_host_register /
 * c3_host_unregister / c3_model_set_lock_sources (hipHostRegister on a whole np.load'ed tensor file or on libclair3's
 * fa_data.matrix buffer, preprocess/CreateTensorFullAlignmentFromCffi.py:136-168).  On ROCm 7.2 a process that registers and
 * unregisters host ranges while ANOTHER HIP user in it (PyTorch) copies from pageable memory dies with "Memory access fault
 * by GPU" sooner or later (tests/diag/register_vs_torch_probe.py shows it with hipHostRegister and torch alone), and a C ABI
 * cannot see who shares its process: the entries were retired in round 6.  A foreign buffer -- numpy, a memory-mapped tensor
 * file, libclair3's matrix -- is handed to c3_predict / c3_predict_submit as it is and staged through the library's own
 * pinned memory (kept out of forked children); x_host may be reused as soon as the call returns. */
/* device-resident forward: x_dev / y_dev are device pointers on the model's device, stream is a
 * hipStream_t (NULL = the HIP null stream, i.e. PyTorch's default stream).  Asynchronous with respect to the
 * host; ordered like any other work on that stream.  Calls on one handle must not overlap each other (one
 * workspace per handle): keep several batches in flight with several handles. */
int c3_predict_device(c3_model *m, const void *x_dev, int x_dtype, int64_t batch, float *y_dev, void *stream);
/* The same forward with the range guard of c3_predict_wait (the device-resident entry a sharded job uses,
 * clair3_amd/dist.py): after the kernels it scans the rows for non-finite values on the device, reads the range flag,
 * and -- should either be raised -- switches the handle to the fp32 matrix instructions and runs the batch again.
 * Synchronises `stream` before returning (that is the price of the check; c3_predict_device stays asynchronous). */
int c3_predict_device_checked(c3_model *m, const void *x_dev, int x_dtype, int64_t batch, float *y_dev, void *stream);
/* For callers of the unchecked entry: *flag_out != 0 when any fp16x3 batch of this handle raised the range flag
 * (bit 0: a convolution output reached 16000, bit 1: a non-finite row seen by the checked entry), *on_fp32_out != 0 when
 * the handle has been switched to the fp32 matrix instructions.  Synchronises the device. */
int c3_model_range_status(c3_model *m, int *flag_out, int *on_fp32_out);
/* Pileup only: windows gathered on the device out of ONE region matrix instead of `batch` pre-sliced
 * copies.  region_host: (n_cols, C) int8|int32|int64 counts exactly as calculate_clair3_pileup returns them for a region
 * (src/clair3_pileup.h:113; preprocess/CreateTensorPileupFromCffi.py:143-146) -- C3_DTYPE_I64 is plp_data.matrix itself
 * (size_t), narrowed to int32 while it is staged, so the numpy copy of :143-146 is not needed either; starts_host[b] = first column of window
 * b, i.e. the `offset` the reference slices at (CreateTensorPileupFromCffi.py:362-364: result[0][offset:offset+33]).
 * Equivalent to c3_predict on the sliced windows, bit for bit; candidate filtering stays with the caller. */
int c3_predict_pileup_region(c3_model *m, const void *region_host, int x_dtype, int64_t n_cols, const int32_t *starts_host,
                             int64_t batch, float *y_host);
/* SURVEY 8f N1 (first slice): the arithmetic of the reference decoder, clair3/CallVariants.py:510-659
 * (possible_outcome_probabilites_from).  For every probability row y_host[b] (24 or 90 floats, as produced by
 * c3_predict) and the gt21 index of its reference base pair ref21_host[b] (0 AA, 4 CC, 7 GG, 9 TT -- reference_gt21 at
 * :520,:570), computes for the ten outcome classes in the order of the reference's max(...) call (:722-733: homo_Ref,
 * homo_SNP, hetero_SNP, homo_Ins, homo_Del, hetero_ACGT_Ins, hetero_InsIns, hetero_ACGT_Del, hetero_DelDel,
 * hetero_InsDel) the maximum of the class's probability list and the position of its first occurrence in the
 * reference's enumeration order, plus early_host[b] = 1 where the reference takes the homo-reference early exit
 * (:532-534, :573-576).  Products are float32 in the reference's multiplication order: the values are bit-identical
 * to the numpy float32 scalars of the reference, so `maximum_probability in <class list>` (:741-749) equals
 * `maxp[b][class] == max over classes`.  Allele strings, alt_info and the retry loop stay in Python.
 * maxp_host: [batch][10] float, argmax_host: [batch][10] int32, early_host: [batch] bytes.  Synchronous. */
int c3_outcome_maxima(c3_model *m, const float *y_host, int64_t batch, const uint8_t *ref21_host, float *maxp_host,
                      int32_t *argmax_host, uint8_t *early_host);
/* The decoder columns for rows that are already on the host: rows_host[batch][output_size + C3_DECODE_COLS] receives
 * y_host[b] followed by its columns (same kernel as the c3_model_set_decode_columns path).  Synchronous. */
int c3_decode_columns(c3_model *m, const float *y_host, int64_t batch, float *rows_host);
/* SURVEY 8f N1, the host side of the decoder: the VCF text of the rows of a batch whose first decision stands, in one pass over the
 * batch -- what clair3/CallVariants.py does per row in batch_output (:1069-1116) -> output_with (:1119-1394) -> output_from's first
 * pass (:676-1008) with the allele lookups find_alt_base (:662-673), insertion_ / deletion_bases_using_alt_info_from (:117-201), for
 * rows that carry the decoder columns.  Plain host code (no device, safe in the forked decode workers of
 * clair3/CallVariantsFromCffi.py:302-353).  An accelerator of clair3_amd/vcf_rows.py, not a second decoder: rows that are not
 * plainly the common case get status 1 and are printed by the caller's Python path (the walk over rejected candidates, shared
 * maxima, odd bytes).
 *   cfg         what the reference's OutputConfig / param say: width 24|90, flank = param.flankingBaseNum, show_reference,
 *               keep_iupac, has_qs_pass / qs_pass = quality_score_for_pass, pileup ('P' / 'F' in INFO), max_len = VariantLength.max,
 *               infer = maximum_variant_length_that_need_infer, phred_trans = CallVariants.Phred_Trans, f32_arith = 1 when the
 *               caller's numpy evaluates `1.0 - float32` in float32 (numpy >= 2: quality_score_from :375-381 depends on it),
 *               gt[4] = genotype_string_from of homo_reference, homo_variant, hetero_variant, hetero_variant_multi; walk = 1: rows
 *               whose first candidate the reads do not offer are walked here too (the later passes of output_from's loop), 0: handed back
 *   pos_text / alt_text   the n chr_pos_seq / alt_info strings, NUL-separated
 *   rows        n rows of row_stride_floats floats: width probabilities + C3_DECODE_COLS columns
 *   out, out_off[n + 1], status[n]   text of row i = out[out_off[i] .. out_off[i + 1]) when status[i] is 0 (first decision) or 2 (after
 *               rejected candidates); empty: the reference prints nothing for it.  status[i] == 1: the caller prints the row itself */
typedef struct {
    int32_t width, flank, show_reference, keep_iupac, has_qs_pass, pileup, max_len, infer, f32_arith, walk;
    int32_t gvcf, haploid;  /* gvcf: rows carry the PL field (output_config.gvcf, clair3/CallVariants.py:1360-1378, compute_PL :1397-1454);
                             * haploid: bit 0 is_haploid_precise_mode_enabled, bit 1 is_haploid_sensitive_mode_enabled (:1191-1199, :1327-1329) */
    int32_t long_indel, long_infer;  /* --enable_long_indel (and not param.cal_precise_long_indel_af): the reads of insertion alleles within long_prop of a
                                      * long allele's length count with it (get_long_indel_read_count, clair3/CallVariants.py:383-402); long_infer =
                                      * param.maximum_variant_length_that_need_infer (50), long_prop = param.long_indel_distance_proportion (0.1) */
    double qs_pass, phred_trans, long_prop;
    char gt[4][8];
} c3_rows_config;
int c3_vcf_rows(const c3_rows_config *cfg, int64_t n, const char *pos_text, int64_t pos_bytes, const char *alt_text, int64_t alt_bytes,
                const float *rows, int64_t row_stride_floats, char *out, int64_t out_cap, int64_t *out_off, uint8_t *status);
/* ---- the collective of the sharded job ----
 * The reference meets its per-GPU workers on disk (one VCF shard per worker, merged by SortVcf:
 * clair3/CallVariantsFromCffiGPU.py:138-199, preprocess/SortVcf.py:290-362).  Here the probability rows of every
 * rank's window shard travel to one rank in a single gather(v), issued DIRECTLY on RCCL (grouped ncclSend / ncclRecv
 * over xGMI, librccl bound with dlopen) on the caller's HIP stream, ordered behind the forward pass on that stream.
 *   c3_comm_unique_id   rank 0 fills 128 bytes (ncclGetUniqueId); the caller hands them to every rank (any control
 *                       plane: a file, the launcher's store) -- rendezvous is not this library's business
 *   c3_comm_create      every rank, same id; world == 1 needs no id and no RCCL (the gather is a device copy) -- unless the
 *                       environment says C3HIP_FORCE_RCCL=1 (test knob): then a world of one makes its own id and a one-rank
 *                       communicator (ncclCommInitRank, nranks = 1) and its gather is a grouped self ncclSend / ncclRecv, so
 *                       the whole call sequence meets the real librccl on a one-GPU box (tests/test_comm_gpu.py)
 *   c3_gather_rows      rows_dev: this rank's counts[rank] rows of row_floats floats (device); counts: rows of every rank
 *                       (host, identical on all ranks -- they follow from the shard ranges); all_dev (rank dst only):
 *                       sum(counts) rows, rank-major = window order for contiguous shards.  Asynchronous on `stream`. */
typedef struct c3_comm c3_comm;
int c3_comm_unique_id(void *id128);
c3_comm *c3_comm_create(const void *id128, int rank, int world, int device);
int c3_comm_destroy(c3_comm *c);
int c3_gather_rows(c3_comm *c, const float *rows_dev, int row_floats, const int64_t *counts, float *all_dev, int dst, void *stream);
/* what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank): *ranks_out ranks, this one *rank_out
 * (may be NULL); world == 1 answers without RCCL.  bench.py prints it as rccl_ranks_seen. */
int c3_comm_count(c3_comm *c, int *ranks_out, int *rank_out);
/* give up on a collective that does not complete (ncclCommAbort); from then on c3_gather_rows on this handle is the local
 * copy of THIS rank's rows (counts[] stays the caller's array over the original ranks; all_dev is required) and the caller
 * routes its rows another way (clair3_amd/dist.py falls back to torch.distributed) */
int c3_comm_abort(c3_comm *c);
/* watchdog for work queued on `stream` of `device` (hipStreamQuery polled every 50 us): 0 = finished, 1 = still running after
 * timeout_ms (c3_last_error() == "timeout"; timeout_ms < 0 waits for ever), other = error */
int c3_stream_wait(void *stream, int device, int timeout_ms);
/* A hint, not a requirement: how many handles the CALLER keeps feeding this GPU side by side (default 1 -- the worker of
 * clair3/CallVariantsFromCffiGPU.py:163-199 after callvar.install(): one process, one handle per GPU).  Tile shapes of the pileup
 * kernels follow it (alone: 8-window LSTM tiles and 240 projection workgroups so that a 1024-window batch reaches every CU;
 * sharing: 16-window tiles and 120 workgroups, because the other batches fill the rest).  Rows are bit-identical either way. */
int c3_model_set_sharing(c3_model *m, int handles);
/* which kernel forms the handle's last forward pass took, as "key=value ..." text; bench.py reports it next to its rates */
int c3_model_describe(c3_model *m, char *buf, int buf_bytes);
/* blocks until everything enqueued on the model's own stream has finished */
int c3_model_synchronize(c3_model *m);
int c3_model_destroy(c3_model *m);

/* ---- introspection used by the parity tests and bench.py (not needed by a pipeline) ---- */

/* Copy an intermediate activation of the most recent predict call (its last micro-batch) to the host.
 * names: pileup  "lstm1_out" (B,33,256) "lstm2_out" (B,33,320) "l4_out" (B,128)
 *        full-aln "act0".."act8" NHWC conv outputs, "spp" (B,3584), "l4_out" (B,256).
 * n_floats must equal the tensor's element count for the last batch. */
int c3_debug_fetch(c3_model *m, const char *name, float *host_out, int64_t n_floats);
/* enable=1: every layer writes to its own buffer (otherwise three buffers are recycled and only the
 * non-recycled tensors can be fetched).  Also enabled by the environment variable C3HIP_KEEP_ACTIVATIONS. */
int c3_debug_keep_activations(c3_model *m, int enable);

/* Per-kernel-family HIP-event timing on the launch stream.  enable=1 brackets every kernel launch with
 * hipEvents (small overhead -- keep it off when measuring whole-job throughput). */
int c3_profile_enable(c3_model *m, int enable);
int c3_profile_reset(c3_model *m);
/* Fills up to max_entries records; returns the number of families, <0 on error.
 * flops are ALGORITHMIC (2*MACs of the reference layer shapes, no padding). */
typedef struct {
    char name[48];
    int64_t launches;
    double total_ms;
    double flops; /* summed over the recorded launches */
    double bytes; /* algorithmic bytes moved (inputs+outputs+weights once per launch) */
    double mfma_flops;       /* FLOP the matrix instructions EXECUTED (tile padding, fp16x3 piece products, Winograd reduction) */
    double mfma_peak_tflops; /* dense peak of the matrix instruction the family issues: 2500 (16-bit inputs) or 157.3 (fp32 inputs); 0 = no matrix work */
} c3_kernel_stat;
int c3_profile_read(c3_model *m, c3_kernel_stat *out, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* C3HIP_H */
