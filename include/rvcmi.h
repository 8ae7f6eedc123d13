/*
 * rvcmi.h -- C ABI of the MI355X-native RVC hot path (librvcmi.so).
 *
 * The reference (fumiama/Retrieval-based-Voice-Conversion-WebUI) has no C ABI of its own: its
 * boundary for this path is two Python objects,
 *
 *   (1) the faiss index held by the pipeline
 *         faiss.read_index(path)                infer/modules/vc/pipeline.py:214, infer/lib/rtrvc.py:56
 *         index.reconstruct_n(0, index.ntotal)  infer/modules/vc/pipeline.py:215, infer/lib/rtrvc.py:57
 *         index.search(npy, k=8)                infer/modules/vc/pipeline.py:126, infer/lib/rtrvc.py:172
 *         the (1/score)^2 blend                 infer/modules/vc/pipeline.py:129-138
 *   (2) the generator module `net_g.dec`
 *         NSFGenerator.forward(x, f0, g, n_res) rvc/layers/nsf.py:145-191
 *         Generator.forward(x, g, n_res)        rvc/layers/generators.py:70-98
 *         built / weight-norm-folded by         rvc/synthesizer.py:10-28
 *
 * Each entry point below names the reference call it stands in for.  The Python mirror of those
 * objects (package `retrieval-based-voice-conversion-webui_amd`, imported as `rvc_amd`) binds this
 * header with ctypes; INTEGRATION.md shows the two-line patch on the reference side.
 *
 * Conventions
 *   - extern "C", opaque handles, plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success and a negative rvcmi_status on failure; the message is
 *     available from rvcmi_last_error() (thread-local).  Nothing throws across the ABI.
 *   - all `*_dev` pointers are DEVICE pointers owned by the caller (hipMalloc / torch.cuda);
 *     all work is enqueued on the caller's `stream` (a hipStream_t passed as void*); no hidden
 *     synchronisation, no allocation after create  =>  every forward/search is hipGraph-capturable.
 *   - handles are immutable after create except for their private workspace: one in-flight
 *     forward per handle (use one handle per stream for concurrency).
 */
#ifndef RVCMI_H
#define RVCMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version.  2: rvcmi_nsf_forward takes lengths_dev as its 4th argument (ragged batches).  A binding must compare
 * rvcmi_version() with the RVCMI_VERSION it was written against before its first call: a v1 caller on a v2 library would
 * shift every pointer by one.                                                                                              */
#define RVCMI_VERSION 2

typedef enum {
    RVCMI_OK = 0,
    RVCMI_ERR_INVALID = -1,     /* bad argument / unsupported configuration            */
    RVCMI_ERR_HIP = -2,         /* a HIP runtime call failed                            */
    RVCMI_ERR_IO = -3,          /* file missing / truncated / not an IVF-Flat L2 index  */
    RVCMI_ERR_NOMEM = -4,       /* shape exceeds the handle's max_B / max_T             */
    RVCMI_ERR_MISSING = -5      /* a required weight tensor was not supplied            */
} rvcmi_status;

const char* rvcmi_last_error(void);
int rvcmi_version(void);

/* ------------------------------------------------------------------------------------------- */
/* Generator (NSF-HiFi-GAN)                                                                      */
/* ------------------------------------------------------------------------------------------- */

#define RVCMI_MAX_UPS 8
#define RVCMI_MAX_RB 4
#define RVCMI_MAX_DIL 4

typedef enum {
    RVCMI_OPERAND_F32 = 0,   /* exact-fp32 VALU convolutions (bring-up / highest fidelity) */
    RVCMI_OPERAND_BF16 = 1,  /* bf16 MFMA operands, fp32 accumulate + fp32 residual stream */
    RVCMI_OPERAND_F16 = 2    /* fp16 MFMA operands, fp32 accumulate + fp32 residual stream */
} rvcmi_operand;

/* Mirrors the positional `cpt["config"]` list consumed by rvc/synthesizer.py:10-22
 * (written by infer/lib/train/process_ckpt.py:23-42). */
typedef struct {
    int inter_channels;                       /* 192                                          */
    int upsample_initial_channel;             /* 512                                          */
    int gin_channels;                         /* 256 (0 = no speaker conditioning)            */
    int sr;                                   /* 32000 / 40000 / 48000                        */
    int use_f0;                               /* 1 = NSFGenerator, 0 = Generator              */
    int n_ups;
    int upsample_rates[RVCMI_MAX_UPS];
    int upsample_kernel_sizes[RVCMI_MAX_UPS];
    int n_resblock_kernels;                   /* 3                                            */
    int resblock_kernel_sizes[RVCMI_MAX_RB];
    int n_dilations[RVCMI_MAX_RB];
    int resblock_dilation_sizes[RVCMI_MAX_RB][RVCMI_MAX_DIL];
    int operand;                              /* rvcmi_operand                                */
} rvcmi_nsf_config;

/* One fp32 host tensor; `name` is the state_dict key under "dec." after remove_weight_norm()
 * (e.g. "ups.0.weight", "resblocks.3.convs1.0.bias").                                         */
typedef struct {
    const char* name;
    const float* data;      /* host pointer, C-contiguous                                       */
    int ndim;
    int64_t shape[4];
} rvcmi_tensor;

typedef struct rvcmi_nsf rvcmi_nsf;

/* Stands in for rvc/synthesizer.py:10-28 (construction + weight prep of `net_g.dec`).
 * Packs the weights into MFMA fragment order on `device` and allocates a workspace sized for
 * max_B utterances of max_T frames.                                                            */
int rvcmi_nsf_create(const rvcmi_nsf_config* cfg, const rvcmi_tensor* weights, int n_weights,
                     int device, int max_B, int max_T, rvcmi_nsf** out);
int rvcmi_nsf_destroy(rvcmi_nsf* h);

/* Stands in for NSFGenerator.forward (rvc/layers/nsf.py:145-191) / Generator.forward
 * (rvc/layers/generators.py:70-98).
 *   x_dev      [B, inter, T] fp32 (the reference's channel-first layout)
 *   f0_dev     [B, T] fp32 Hz, 0 = unvoiced; NULL iff use_f0 == 0
 *   g_dev      [B, gin] fp32 speaker embedding (emb_g(sid)), or NULL
 *   noise_dev  [B, T*upp] fp32 N(0,1) -- the draw the reference makes at generators.py:192 --
 *              or NULL for zeros.  (The reference's rand_ini at :164-166 is forced to 0 for the
 *              only harmonic there is, so it never reaches the output.)
 *   n_res      -1 = none; otherwise the realtime "return_length2" resample target, nsf.py:155-162
 *   out_dev    [B, T_out*upp] fp32, T_out = n_res if n_res >= 0 else T
 *   lengths_dev  NULL, or int32 [B] on the device with 1 <= lengths[b] <= T: a RAGGED batch.  Item b is
 *              computed exactly as a separate call with T = lengths[b] on its first lengths[b] frames (lengths[b] * upp
 *              noise samples) would compute it: every layer zero-pads its input behind the item's own last row, so the
 *              segments of a long file (infer/modules/vc/pipeline.py:205-209, 301-343) and the utterances of a folder
 *              (infer/modules/vc/modules.py:201-266 vc_multi) convert in ONE call with the waveform of sequential calls.
 *              Output samples [lengths[b] * upp, T * upp) of item b are zero.  Not combinable with n_res.
 */
int rvcmi_nsf_forward(rvcmi_nsf* h, int B, int T, const int* lengths_dev, const float* x_dev, const float* f0_dev,
                      const float* g_dev, const float* noise_dev, int n_res, float* out_dev,
                      void* stream);

int rvcmi_nsf_upp(const rvcmi_nsf* h);               /* prod(upsample_rates)                    */
size_t rvcmi_nsf_workspace_bytes(const rvcmi_nsf* h);

/* Stage taps for bring-up and per-layer parity tests (not on the product path): runs the forward
 * up to the named activation and copies it to the host in the reference's channel-first layout
 * [B, C, L].  `what`: "har" ([B,1,T*upp]), "pre", "up<i>" (after ups+noise_convs), "stage<i>"
 * (the SUM of the stage's ResBlocks, i.e. before the /num_kernels of nsf.py:186).
 * Synchronises the stream.                                                                       */
int rvcmi_nsf_debug_forward(rvcmi_nsf* h, int B, int T, const float* x_dev, const float* f0_dev,
                            const float* g_dev, const float* noise_dev, int n_res, const char* what,
                            float* out_host, size_t capacity_floats, int64_t shape_out[3],
                            void* stream);

/* Per-kernel HIP-event timing (bench.py's roofline leg).  When enabled, every launch of the
 * forward is bracketed by hipEvents on the caller's stream; not usable under graph capture.
 * rvcmi_nsf_profile_read returns, for kernel class i < *n, its name, launch count, total ms
 * and algorithmic flops/bytes accumulated since the last reset.                                */
typedef struct {
    char name[48];
    int64_t launches;
    double ms;
    double flops;      /* 2*Cin*Cout*k*L summed over launches                                  */
    double bytes;      /* algorithmic bytes moved to/from global memory                        */
} rvcmi_kernel_stat;
int rvcmi_nsf_profile_enable(rvcmi_nsf* h, int enable);
/* Development / test options of ONE handle (kernel-family forcing, tile heights, phase stamps, timing ablations).  A handle
 * reads the environment variables RVCMI_<KEY> exactly once, when it is created; afterwards only this call changes an option
 * (value NaN = back to the default).  The forward / search paths never read the environment.  Keys: see the option comments
 * of the handle structs in csrc/nsf.hip, csrc/rb_stream.hpp, csrc/front.hip, csrc/ivf.hip.  Not part of the reference's
 * surface; product code does not call these. */
int rvcmi_nsf_set_option(rvcmi_nsf* h, const char* key, double value);
int rvcmi_nsf_profile_read(rvcmi_nsf* h, rvcmi_kernel_stat* stats, int capacity, int* n, int reset);

/* ------------------------------------------------------------------------------------------- */
/* Synthesizer front: enc_p + prior sampling + flow^-1              */
/* ------------------------------------------------------------------------------------------- */

/* What SynthesizerTrnMsNSFsid.infer (rvc/layers/synthesizers.py:160-203) runs before self.dec:
 *     m_p, logs_p, x_mask = self.enc_p(phone, pitch, phone_lengths, flow_head)   encoders.py:134-159
 *     z_p = (m_p + exp(logs_p) * randn_like(m_p) * 0.66666) * x_mask             synthesizers.py:182-183
 *     z   = self.flow(z_p, x_mask, g=g, reverse=True)                            residuals.py:319-321
 * With this handle and rvcmi_nsf the whole `infer` runs without a torch op in between.
 * Hyper-parameters are the positional config list of a checkpoint (configs/v2/48k.json:19-27). */
typedef struct {
    int in_channels;          /* 768 (v2, SynthesizerTrnMs768NSFsid) or 256 (v1)                */
    int inter_channels;       /* 192                                                            */
    int hidden_channels;      /* 192                                                            */
    int filter_channels;      /* 768                                                            */
    int n_heads;              /* 2                                                              */
    int n_layers;             /* 6                                                              */
    int kernel_size;          /* FFN kernel, 3                                                  */
    int window_size;          /* relative-attention window, 10 (encoders.py:21)                 */
    int gin_channels;         /* 256                                                            */
    int use_f0;               /* emb_pitch present (encoders.py:107-108)                        */
    int flow_n_flows;         /* 4  (residuals.py:275)                                          */
    int flow_n_layers;        /* 3  (synthesizers.py:111-113)                                   */
    int flow_kernel_size;     /* 5                                                              */
    int flow_dilation_rate;   /* 1                                                              */
    int operand;              /* RVCMI_OPERAND_F16 / _BF16 (MFMA operands; everything else fp32) */
} rvcmi_front_config;

typedef struct rvcmi_front rvcmi_front;

/* Stands in for the enc_p / flow part of get_synthesizer (rvc/synthesizer.py:10-28).  `weights` are fp32 HOST
 * tensors named by the keys of net_g.state_dict() AFTER remove_weight_norm(): "enc_p.emb_phone.weight",
 * "enc_p.encoder.attn_layers.0.conv_q.weight", "enc_p.encoder.attn_layers.0.emb_rel_k", ...,
 * "enc_p.proj.weight", "flow.flows.0.pre.weight", "flow.flows.0.enc.in_layers.0.weight", ...,
 * "flow.flows.6.post.bias".  The channel Flip modules are folded into the packed weights.       */
int rvcmi_front_create(const rvcmi_front_config* cfg, const rvcmi_tensor* weights, int n_weights, int device,
                       int max_B, int max_T, rvcmi_front** out);
int rvcmi_front_destroy(rvcmi_front* h);

/* One pass.  phone_dev [B][T][in_channels] fp32 (features after the x2 interpolation, pipeline.py:146-150);
 * pitch_dev [B][T] int64 coarse bins or NULL (no-f0 models); lengths_dev [B] int64 (phone_lengths) or NULL = all T;
 * g_dev [B][gin] fp32 = emb_g(sid); noise_dev [B][inter][T - flow_head] fp32 standing for randn_like(m_p);
 * flow_head = max(skip_head - 24, 0) for the realtime partial decode (synthesizers.py:175-181), else 0.
 * z_out_dev [B][inter][T - flow_head] fp32 = z * x_mask, the layout rvcmi_nsf_forward takes as x_dev.  */
int rvcmi_front_forward(rvcmi_front* h, int B, int T, const float* phone_dev, const int64_t* pitch_dev,
                        const int64_t* lengths_dev, const float* g_dev, const float* noise_dev, int flow_head,
                        float* z_out_dev, void* stream);
size_t rvcmi_front_workspace_bytes(const rvcmi_front* h);

/* Test hook: stop after an internal stage and copy it to the host, channels-last [B][T'][192].
 * what: "emb", "attn0", "layer0".."layer5" (encoder stream), "z_p", "flow3".."flow0" (flow stream after a coupling). */
int rvcmi_front_debug_forward(rvcmi_front* h, int B, int T, const float* phone_dev, const int64_t* pitch_dev,
                              const int64_t* lengths_dev, const float* g_dev, const float* noise_dev, int flow_head,
                              const char* what, float* out_host, size_t capacity_floats, int64_t shape_out[3],
                              void* stream);
int rvcmi_front_set_option(rvcmi_front* h, const char* key, double value);
int rvcmi_front_profile_enable(rvcmi_front* h, int enable);
int rvcmi_front_profile_read(rvcmi_front* h, rvcmi_kernel_stat* stats, int capacity, int* n, int reset);

/* ------------------------------------------------------------------------------------------- */
/* IVF-Flat retrieval (faiss IndexIVFFlat, METRIC_L2)                                           */
/* ------------------------------------------------------------------------------------------- */

typedef struct rvcmi_ivf rvcmi_ivf;

/* faiss.read_index(path)  (pipeline.py:214).  Parses the IwFl/IxF2/ilar on-disk layout.       */
int rvcmi_ivf_create_from_file(const char* path, int device, rvcmi_ivf** out);
/* index.train(big_npy) + index.add(big_npy)  (web.py:554-563), on the GPU:
 *   k-means for the nlist centroids (niter Lloyd iterations from nlist seeded training vectors; every assignment
 *   is the exact fp64 nearest centroid, computed by the search path's own coarse kernels; empty lists are re-seeded
 *   by splitting the largest one), then every vector goes to the list of its nearest centroid, ids = row numbers
 *   in add order.  x_host [n,d] fp32 HOST (what np.load gives).  nprobe = 1 (web.py:552).  objective_out
 *   (optional, niter+1 doubles) receives the sum of squared distances at every assignment step.
 *   faiss' own k-means (its RNG, sub-sampling and split heuristics) is not reproduced: retrieval semantics do not
 *   depend on how the centroids were found, and the reference pins nothing here.                      */
int rvcmi_ivf_build(int d, int64_t n, const float* x_host, int64_t nlist, int niter, uint64_t seed, int device,
                    double* objective_out, rvcmi_ivf** out);
/* The large-set branch of the index recipe (web.py:522-536: more than 2e5 feature rows are replaced by 10k k-means centres,
 * sklearn MiniBatchKMeans there) without building an index: the SAME Lloyd iterations as rvcmi_ivf_build -- niter updates
 * from k seeded training vectors, exact fp64 assignments, a cluster that loses all its points is re-seeded by splitting the
 * largest one, so all k centres are valid -- and only the centres come back.  x_host [n,d] fp32 HOST, centroids_out_host
 * [k,d] fp32 HOST, objective_out optional (niter doubles).                                                              */
int rvcmi_kmeans(int d, int64_t n, const float* x_host, int64_t k, int niter, uint64_t seed, int device, double* objective_out,
                 float* centroids_out_host);
/* faiss.write_index(index, path)  (web.py:571) -- so indices round-trip with stock RVC.       */
int rvcmi_ivf_write_file(const rvcmi_ivf* h, const char* path);

/* Direct construction from host arrays (what index.train()+index.add() leave behind, web.py:553-563):
 * centroids [nlist,d], list_offsets [nlist+1], ids [n] and vecs [n,d] in list-major order.     */
int rvcmi_ivf_create(int d, int64_t n, int64_t nlist, int nprobe, const float* centroids,
                     const int64_t* list_offsets, const int64_t* ids, const float* vecs,
                     int device, rvcmi_ivf** out);
int rvcmi_ivf_destroy(rvcmi_ivf* h);

int rvcmi_ivf_d(const rvcmi_ivf* h);
int64_t rvcmi_ivf_ntotal(const rvcmi_ivf* h);            /* index.ntotal                         */
int64_t rvcmi_ivf_nlist(const rvcmi_ivf* h);
int rvcmi_ivf_nprobe(const rvcmi_ivf* h);                /* extract_index_ivf(index).nprobe      */
int rvcmi_ivf_set_nprobe(rvcmi_ivf* h, int nprobe);      /* web.py:551-552                       */

/* Pre-size the search workspace for up to max_nq queries (searches with nq above the current
 * reservation grow it, which allocates: reserve before hipGraph capture).                      */
int rvcmi_ivf_reserve(rvcmi_ivf* h, int64_t max_nq);

/* index.search(x, k)  (pipeline.py:126): q_dev [nq,d] fp32 -> D_dev [nq,k] fp32 squared-L2
 * ascending, I_dev [nq,k] int64 (-1 / FLT_MAX padded).  k <= 8.  Distances are evaluated in fp64
 * on the fp32 inputs; ties break to the lowest id.                                             */
int rvcmi_ivf_search(rvcmi_ivf* h, int64_t nq, const float* q_dev, int k, float* D_dev,
                     int64_t* I_dev, void* stream);

/* search + pipeline.py:129-138 fused, everything device-resident:
 *   w = (1/D)^2 / sum; feats = (sum_k w_k * big_npy[I_k]) * index_rate + (1-index_rate) * feats
 * feats_dev [nq,d] fp32 is updated in place.  skip_if_short != 0 reproduces the realtime guard
 * `if (ix >= 0).all()` of infer/lib/rtrvc.py:173 (per call, not per row).                       */
int rvcmi_ivf_search_blend(rvcmi_ivf* h, int64_t nq, float* feats_dev, float index_rate, int k,
                           int skip_if_short, void* stream);

/* rvcmi_ivf_search_blend followed by what Pipeline.vc does next (pipeline.py:140-159), in one pass over the rows:
 *   F.interpolate(scale_factor=2) (nearest: out frame t <- row t/2), truncation to p_len <= 2*nq, and -- when
 *   pitchf_dev is not NULL (the `protect < 0.5` branch) -- feats*pitchff + feats0*(1-pitchff) with
 *   pitchff = pitchf[t] < 1 ? protect : 1.  feats_dev [nq,d] is NOT modified; out_dev is [p_len,d].            */
int rvcmi_ivf_search_blend_expand(rvcmi_ivf* h, int64_t nq, const float* feats_dev, float index_rate, int k,
                                  int skip_if_short, const float* pitchf_dev, float protect, int64_t p_len,
                                  float* out_dev, void* stream);

/* index.reconstruct_n(i0, n) -> out_host [n,d] rows in id order (pipeline.py:215).             */
int rvcmi_ivf_reconstruct_n(const rvcmi_ivf* h, int64_t i0, int64_t n, float* out_host);

/* The coarse centroids [nlist, d] (k-means cluster centres of a built index), e.g. as the 10k-centre reduction of a
 * large training set (web.py:522-536).                                                                             */
int rvcmi_ivf_centroids(const rvcmi_ivf* h, float* out_host);

/* Copies the packed index blob into caller-owned device memory (>= the size rvcmi_ivf_blob reports) on `stream`:
 * the source buffer of the RCCL broadcast of rvc_amd.dist.broadcast_index, made by this library's own HIP runtime.  */
int rvcmi_ivf_blob_copy(const rvcmi_ivf* h, void* dst_dev, size_t capacity, void* stream);

/* The whole index as ONE device blob (header + centroids + offsets + ids + vectors) so that a
 * single RCCL broadcast replicates it across the GPUs of a node.                */
int rvcmi_ivf_blob(const rvcmi_ivf* h, void** dev_ptr, size_t* bytes);
int rvcmi_ivf_create_from_blob(void* dev_ptr, size_t bytes, int device, int take_ownership,
                               rvcmi_ivf** out);

/* Per-phase HIP-event timing for bench.py (coarse / scan / blend), same contract as the nsf one. */
int rvcmi_ivf_set_option(rvcmi_ivf* h, const char* key, double value);
int rvcmi_ivf_profile_enable(rvcmi_ivf* h, int enable);
int rvcmi_ivf_profile_read(rvcmi_ivf* h, rvcmi_kernel_stat* stats, int capacity, int* n, int reset);

/* ------------------------------------------------------------------------------------------- */
/* Device-resident glue: stateless, everything on the caller's stream */
/* ------------------------------------------------------------------------------------------- */

/* The x2 interpolation + protect mix of pipeline.py:140-159 when NO index is used (index_rate == 0):
 * out[t] = feats[t/reps] (* pf + feats[t/reps] * (1 - pf) when pitchf_dev != NULL).                */
int rvcmi_glue_expand_protect(const float* feats_dev, int64_t nq, int d, int reps, const float* pitchf_dev,
                              float protect, int64_t p_len, float* out_dev, void* stream);

/* RMVPE salience [n,nbins=360] -> what Generator.calculate(..., "rmvpe") returns (rvc/f0/gen.py:43-123):
 *   _to_local_average_cents + _decode (rvc/f0/rmvpe.py:119-164), _resize_f0 to p_len and _interpolate_f0
 *   (rvc/f0/f0.py:31-78), post_process (rvc/f0/gen.py:10-41): key shift 2^(f0_up_key/12), mel binning to 1..255.
 * fp64 throughout, like numpy.  scratch_dev: n doubles.  pitch_dev [p_len] int64, pitchf_dev [p_len] fp32.
 * Any length: the single sequential pass keeps its work arrays in LDS up to n + p_len = 20480 frames and
 * in scratch_dev / pitch_dev beyond that (the reference computes f0 once per file, pipeline.py:260-266). */
int rvcmi_glue_rmvpe_f0(const float* salience_dev, int n, int nbins, float thred, int p_len, int f0_up_key,
                        double* scratch_dev, int64_t* pitch_dev, float* pitchf_dev, void* stream);
/* post_process only (f0 in Hz from any other estimator, fp64 [n]).                                 */
int rvcmi_glue_f0_post(const double* f0_dev, int n, int f0_up_key, int64_t* pitch_dev, float* pitchf_dev,
                       void* stream);

/* change_rms (infer/modules/vc/pipeline.py:26-46, called at :351 when rms_mix_rate != 1): mixes the loudness envelope of the
 * input (data1 at sr1 = 16000) into the converted audio data2 (sr2 = tgt_sr), IN PLACE on data2:
 *   data2 *= rms1^(1-rate) * max(rms2,1e-6)^(rate-1), rms_i = half-second frame RMS (librosa.feature.rms, centred, zero
 *   padded) linearly interpolated to len(data2).  scratch_dev: (1 + n1/(sr1/2)) + (1 + n2/(sr2/2)) floats.
 * The frame energies are summed in fp64; librosa sums in float32 in an order nothing in the reference pins.          */
int rvcmi_glue_change_rms(const float* data1_dev, int64_t n1, int sr1, float* data2_dev, int64_t n2, int sr2, float rate,
                          float* scratch_dev, void* stream);

/* pipeline.py:355-359: audio *= 32768 / max(1, abs(audio).max()/0.99), in place.  scratch_dev: 256 floats. */
int rvcmi_glue_scale_int16_range(float* audio_dev, int64_t n, float* scratch_dev, void* stream);

/* SOLA chunk stitching of the realtime path (gui.py:1057-1090): normalised cross-correlation
 * of infer_wav[: Lb + Ls] with sola_buffer [Lb] over offsets 0..Ls, argmax (first maximum), cut, cross-fade with the
 * sin^2 windows, out_block_dev [block_frame] <- result, sola_buffer_dev <- the next tail (in place).
 * offset_out_dev (optional) receives the chosen offset.  Needs Ls + block_frame + Lb <= n.            */
int rvcmi_glue_sola(const float* infer_wav_dev, int64_t n, float* sola_buffer_dev, int Lb, int Ls,
                    const float* fade_in_dev, const float* fade_out_dev, int block_frame, float* out_block_dev,
                    int* offset_out_dev, void* stream);

/* The formant-shift resample of the realtime path (rtrvc.py:248-259, torchaudio.transforms.Resample(orig_freq = upp_res,
 * new_freq = tgt_sr / 100)): out[j * new + p] = sum_{k < K} kernel[p][k] * xpad[j * orig + k], xpad = x with `width` zeros in
 * front and zeros behind; orig / new already divided by their gcd; kernel_dev [new][K] (K = 2 * width + orig) is torchaudio's
 * windowed-sinc table (rvc_amd.realtime.sinc_resample_kernel restates its published formula: hann window, lowpass_filter_width 6,
 * rolloff 0.99).  n_out = ceil(new * n / orig) for the whole signal.  PARITY UNPINNED: torchaudio is not installable offline. */
int rvcmi_glue_resample_poly(const float* x_dev, int64_t n, const float* kernel_dev, int orig, int new_, int K, int width,
                             float* out_dev, int64_t n_out, void* stream);

/* ---- beyond SURVEY.md section 8: the recurrent layer of the RMVPE f0 network -----------------------------------------------
 * Stands in for the `nn.GRU(384, 256, num_layers=1, batch_first=True, bidirectional=True)` of rvc/f0/e2e.py:50-67 (E2E.BiGRU, the
 * first module of E2E.fc), which bench.py --e2e measured at 75-90 % of a whole conversion on PyTorch-ROCm / MIOpen.
 * Weights in torch's layout and gate order (r, z, n), forward direction first: w_ih [2][3H][I], w_hh [2][3H][H], b_ih / b_hh [2][3H],
 * fp32 on the host.  hidden_size must be 256, input_size a multiple of 16 (anything else: RVCMI_ERR_INVALID, the caller keeps torch's).
 * Operands fp16 (x, W_ih, W_hh and the broadcast copy of h), accumulation / gates / state fp32.                                  */
typedef struct rvcmi_gru rvcmi_gru;
int rvcmi_gru_create(int input_size, int hidden_size, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                     int device, rvcmi_gru** out);
int rvcmi_gru_destroy(rvcmi_gru* h);
/* x16_dev [B][T][I] fp16; y_dev [B][T][2H] fp32 = nn.GRU's `output` with h_0 = 0; hn_dev [2][B][H] fp32 = its `h_n` (or NULL).
 * The projection workspace grows with the largest B * T seen (a hipMalloc on such a call: not for use inside a stream capture). */
int rvcmi_gru_forward(rvcmi_gru* h, int B, int T, const void* x16_dev, float* y_dev, float* hn_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RVCMI_H */
