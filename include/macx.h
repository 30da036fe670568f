/* macx.h -- C ABI of libmacx.so: the MI355X-native MAC reasoning cell.
 *
 * The reference (stanfordnlp/mac-network) has no FFI: its hot path is the Python class
 * MACCell (mac_cell.py:30-592) driven by MACnet.MACnetwork (model.py:428-489), and all
 * arithmetic lives in the TensorFlow-1.x runtime.  This header is the boundary a maintainer
 * would bind to replace that class's arithmetic (ctypes stub in INTEGRATION.md).  Every entry
 * point names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; all tensors fp32, row-major, contiguous, 16-byte aligned;
 *     question lengths int32.  All pointers are DEVICE pointers unless marked host.
 *   - ownership: the caller owns every buffer (parameters, inputs, `saved`, `ws`, gradients).
 *     The library never allocates, frees or retains device memory.
 *   - asynchronous: work is enqueued on `stream`; nothing synchronises the device.
 *   - errors: 0 on success, a negative MACX_E* code for a rejected call, or the positive
 *     hipError_t of a failed launch.  No exceptions, no abort().
 *   - determinism: no floating-point atomics anywhere; weight gradients use fixed-order
 *     slab reductions, so two runs on the same inputs are bit-identical.
 */
#ifndef MACX_H
#define MACX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MACX_ABI_VERSION 5

enum {
  MACX_OK = 0,
  MACX_EINVAL = -1,      /* malformed shapes / null pointer / misaligned pointer            */
  MACX_EUNSUPPORTED = -2,/* legal reference option combination without a HIP path yet       */
  MACX_EREJECTED = -3,   /* option value that raises in the reference   */
  MACX_ESMALL = -4       /* `saved` or `ws` smaller than macx_saved_floats / macx_ws_floats */
};

/* activation codes (ops.py:181-187; "RELU" resolves through config.relu, ops.py:161-179) */
enum { MACX_ACT_NON = 0, MACX_ACT_TANH = 1, MACX_ACT_SIGMOID = 2, MACX_ACT_ELU = 3, MACX_ACT_RELU = 4 };
/* state initialisation (mac_cell.py:496-505) */
enum { MACX_INIT_PRM = 0, MACX_INIT_ZERO = 1, MACX_INIT_Q = 2 };
/* write-unit inputs (mac_cell.py:333-340) */
enum { MACX_WRITE_MEM = 0, MACX_WRITE_INFO = 1, MACX_WRITE_SUM = 2, MACX_WRITE_BOTH = 3 };

/* Frozen option surface: the subset of config.py flags (config.py:194-223, 292-387) the cell
 * branches on, resolved to integers by the host (mac-network_amd/options.py). */
typedef struct macx_opts {
  int32_t abi_version;              /* MACX_ABI_VERSION                                        */
  int32_t init_ctrl, init_mem;      /* --initCtrl / --initMem                                  */
  int32_t control_input_unshared;   /* --controlInputUnshared     (mac_cell.py:430-432)        */
  int32_t control_input_act;        /* --controlInputAct          (mac_cell.py:445)            */
  int32_t control_feed_prev;        /* --controlFeedPrev          (mac_cell.py:142)            */
  int32_t control_feed_prev_att;    /* --controlFeedPrevAtt       (mac_cell.py:143)            */
  int32_t control_feed_inputs;      /* --controlFeedInputs        (mac_cell.py:144-146)        */
  int32_t control_cont_act;         /* --controlContAct           (mac_cell.py:149-150)        */
  int32_t read_mem_act;             /* --readMemAct resolved      (mac_cell.py:237)            */
  int32_t read_ctrl_act;            /* --readCtrlAct resolved     (mac_cell.py:262)            */
  int32_t write_inputs;             /* --writeInputs              (mac_cell.py:333-340)        */
  int32_t write_self_att;           /* --writeSelfAtt             (mac_cell.py:316-330)        */
  int32_t write_self_att_cont;      /* --writeSelfAttMod == CONT  (mac_cell.py:318-319)        */
  int32_t write_mem_act;            /* --writeMemAct resolved     (mac_cell.py:355)            */
  int32_t write_gate;               /* --writeGate                (mac_cell.py:358-367)        */
  int32_t write_gate_shared;        /* --writeGateShared          (mac_cell.py:360-361)        */
  float   write_gate_bias;          /* --writeGateBias            (mac_cell.py:363)            */
  int32_t memory_variational_dropout; /* --memoryVariationalDropout (mac_cell.py:214-217)      */
  int32_t gemm_family;              /* kernel family of the knowledge-base GEMMs for EVERY call made with these options
                                       (sizing, begin, step, backward): 0 = the process default (macx_gemm_mode),
                                       1 + MACX_GEMM_NATIVE / _SPLIT / _H2 = that family.  Per call and per thread: two
                                       cells of one process can run on different families side by side               */
  int32_t tune[16];                 /* PER-CALL tuning table (ABI 5; replaces the process-global `macx_debug_set` of ABI <= 4): the A/B
                                       hook of each live kernel-selection decision and the phase masks of the profiling tools.
                                       All zero = the shipped configuration.  tune[k] = 0: the default of key k; else the value
                                       v is passed as v + 1 (MACX_TUNE(v)).  Read when a call is enqueued, from the opts of THAT
                                       call (forward and backward of one run must see the same table): no state lives in the
                                       library, calls on different threads or with different opts never see each other's table.
                                       Keys: MACX_TUNE_* below.  Never needed for correct results. */
} macx_opts;
#define MACX_TUNE(v) ((v) + 1)
enum {
  MACX_TUNE_NATIVE_WAVES = 0, /* waves per workgroup of the NATIVE knowledge-base GEMM: 4 | 8 (default)                        */
  MACX_TUNE_PHASE_MASK = 1,   /* bit mask of TIMING experiments (results are wrong under a non-zero mask; tools/mask_steps.py):
                                 1 skip the GEMM epilogue, 2 skip the in-loop staging, 8 epilogue without its global stores,
                                 32 / 64 re-read K-slice 0 of A / of the weights (cache-hot), 128 weight-gradient contractions on
                                 the f32 TN kernel, 256 S_b kernel on the f32 kernel, 512 / 1024 / 2048 the deferred contractions
                                 without their folds / products / in-loop DMA, bits 12-16 stop chain_fwd after a stage, bits
                                 17-21 chain_bwd                                                                              */
  MACX_TUNE_ROW_TILES = 2,    /* forced row tiles per GEMM workgroup (0 = automatic | 1 | 2 | 4 | 7 | 13)                      */
  MACX_TUNE_CHAIN = 4,        /* 0: the read unit's products as separate launches (what also runs for d > 512 or N < 16);
                                 1 (default): the chain kernels of macx_chain_h2.hip.h                                        */
  MACX_TUNE_SB_DEFER = 5,     /* 0: the per-question contraction S_b = X_b^T dI1_b once per step (it then also delivers dy; what
                                 also runs for N < 32); 1 (default): dy from the chain kernel, S_b of all steps in one launch   */
  MACX_TUNE_CHAIN_KV = 7,     /* K loop of the d = 512 chain kernels: 0 activation fragments requested in front of a slice's
                                 products; -1 (default): in mid-slice                                                         */
  MACX_TUNE_SB_WIDE = 8,      /* dW1a / dW1b: 0 the 128 x 128 per-question S_b kernel; 1 (default) the 128 x 256 one            */
  MACX_TUNE_WGRAD_PIPE = 10,  /* wgrad_h2_kernel<2,2>: 0 a stage requested one iteration ahead; 1 a buffer's halves re-requested
                                 inside the iteration; 2 = 1 + dW2 and dWx as ONE launch; 3 (default) = 2 with half the splits  */
  MACX_TUNE_SB_CONT = 13,     /* sb_h2w_kernel: 1 (default) one stage stream over all steps; 0 drained and re-primed per step    */
  MACX_TUNE_DKB_UNI = 14,     /* merged dKB launch: 1 (default) one fold of the block accumulator per step; 0 per 128-wide K block */
  MACX_TUNE_PRE_FILL = 3,     /* what the filler workgroups of a d = 512 chain_fwd launch do on the CUs its tile grid leaves idle (runs
                                 of macx_cell_forward; 196 tiles on 256 CUs at B = 64): 1 (default) the previous step's write-unit
                                 linear, this step's y = md Wy + by, and stage 0 (dropout(KB) -> fp16 planes, keep bits) of the next
                                 step; 2 the same without the write unit; 0 no fillers: every piece is a launch / a stage of its own */
  MACX_TUNE_DKB_FILL = 15     /* dKB on the CUs a d = 512 chain_bwd launch leaves idle (196 tiles on 256 CUs at B = 64): jobs per
                                 filler workgroup, default 3; 0: all of dKB in the merged launch after the last step            */
};

typedef struct macx_shapes {
  int32_t B;     /* questions in this (shard of the) batch                                     */
  int32_t S;     /* padded question length                                                     */
  int32_t N;     /* knowledge-base cells per question (H*W, model.py:202)                      */
  int32_t d;     /* memDim == ctrlDim == attDim (config.py:294-296); multiple of 128           */
  int32_t p;     /* netLength (config.py:292)                                                  */
  int32_t b0;    /* global index of question 0: data-parallel shard offset (model.py:139-149)  */
  int32_t d_logical; /* 0 (or d): the cell is d wide.  Else: the caller runs a d_logical-wide cell (config.py:294-296 takes
                      * any width) zero-padded to d columns -- weights, biases, inputs -- and the dropout element indices are
                      * taken at the LOGICAL width ((row * d_logical + column): the masks of the unpadded cell); d_logical % 8
                      * == 0, d - 128 < d_logical < d.  The padded columns of every state and gradient are exact zeros. */
} macx_shapes;

/* Dropout of one cell run.  keep == 1 (evaluation, model.py:118-125) is the exact identity. */
typedef struct macx_dropout {
  float keep_memory, keep_read, keep_write;   /* config.py:213-215                              */
  uint32_t seed;                              /* stateless mask stream, see macx_dropout_mask  */
  const uint32_t* mask_word;                  /* NULL, or ONE 32-bit word in DEVICE memory that every dropout site of the run
                                               * XORs into its key when the kernel RUNS (not when it is enqueued):
                                               *   mask(seed, word, site, step, i) = stream keyed site_key(seed, site, step) ^ word.
                                               * The seed travels by value in kernel arguments, so a captured HIP graph would
                                               * replay the masks of the captured seed for ever; the word is read on every
                                               * replay, so ONE capture of a training step draws fresh masks per replay
                                               * (write the word between replays -- e.g. a hash of the iteration number).
                                               * Forward and backward of one run must see the same word.  NULL == word 0 ==
                                               * the masks of ABI <= 3. */
} macx_dropout;

/* Parameters, keyed by the reference's variable names under macModel/MACnetwork/ (SURVEY 8b).
 * Matrices are [in, out] row-major exactly as tf.get_variable stores them. */
typedef struct macx_params {
  const float* initMem;        /* initMem [d]                         (mac_cell.py:498)         */
  const float* initCtrl;       /* initCtrl [d]           (PRM only)                             */
  const float* qInput_W;       /* MACCell/linearLayerqInput/weights/weight [d,d]                */
  const float* qInput_b;       /*                       .../biases/bias [d]                     */
  const float* qInputU_W;      /* linearLayerqInput{i} stacked [p,d,d] (or [1,d,d] if shared)   */
  const float* qInputU_b;      /* [p,d] / [1,d]                                                 */
  const float* ctrlLogits_w;   /* control/inter2logits/linearLayerlogits/weights/weight [d]     */
  const float* ctrlLogits_b;   /* .../biases/bias []                                            */
  const float* contControl_W;  /* control/linearLayercontControl [d or 2d, d]   (args1)         */
  const float* contControl_b;
  const float* contControl2_W; /* .../linearLayercontControl_2 [d,d]   (controlContAct != NON)  */
  const float* contControl2_b;
  const float* projX_W;        /* read/mulmemInter/linearLayerprojX [d,d]                       */
  const float* projX_b;
  const float* projY_W;        /* read/mulmemInter/linearLayerprojY [d,d]                       */
  const float* projY_b;
  const float* memKbProj_W;    /* read/linearLayermemKbProj [2d,d]: rows [0,d) x*y, [d,2d) x    */
  const float* memKbProj_b;
  const float* memKbProj2_W;   /* read/linearLayermemKbProj/linearLayermemKbProj_2 [d,d]        */
  const float* memKbProj2_b;
  const float* kbLogits_w;     /* read/inter2att/inter2logits/linearLayerlogits [d]             */
  const float* kbLogits_b;
  const float* newMemory_W;    /* write/linearLayernewMemory [2d or 3d, d]                      */
  const float* newMemory_b;
  const float* selfCtrl_W;     /* write/linearLayerctrlProj [d,d]               (args3)         */
  const float* selfCtrl_b;
  const float* selfLogits_w;   /* write/inter2attselfAttention/.../linearLayerlogits [d]        */
  const float* selfLogits_b;
  const float* gate_W;         /* write/linearLayergate [d, d or 1]             (args4)         */
  const float* gate_b;
} macx_params;

/* Gradients: same fields, written (not accumulated) by macx_cell_backward. */
typedef struct macx_param_grads {
  float* initMem; float* initCtrl;
  float* qInput_W; float* qInput_b; float* qInputU_W; float* qInputU_b;
  float* ctrlLogits_w; float* ctrlLogits_b;
  float* contControl_W; float* contControl_b; float* contControl2_W; float* contControl2_b;
  float* projX_W; float* projX_b; float* projY_W; float* projY_b;
  float* memKbProj_W; float* memKbProj_b; float* memKbProj2_W; float* memKbProj2_b;
  float* kbLogits_w; float* kbLogits_b;
  float* newMemory_W; float* newMemory_b;
  float* selfCtrl_W; float* selfCtrl_b; float* selfLogits_w; float* selfLogits_b;
  float* gate_W; float* gate_b;
} macx_param_grads;

/* Cell inputs: the tensors MACCell.__init__ stores (mac_cell.py:59-79). */
typedef struct macx_inputs {
  const float* vecQuestions;     /* [B,d]                                                        */
  const float* words;            /* [B,S,d]  questionCntxWords or questionWords (mac_cell.py:570) */
  const int32_t* questionLengths;/* [B]                                                          */
  const float* knowledgeBase;    /* [B,N,d]  stem output (model.py:202)                          */
} macx_inputs;

typedef struct macx_input_grads {
  float* vecQuestions;           /* [B,d]                                                        */
  float* words;                  /* [B,S,d]                                                      */
  float* knowledgeBase;          /* [B,N,d]                                                      */
} macx_input_grads;

/* Segments of the `saved` buffer that the host may view (offsets in floats). */
enum {
  MACX_SEG_CONTROLS = 0,   /* [p+1,B,d]  controls history, entry 0 = initial (mac_cell.py:549,472) */
  MACX_SEG_MEMORIES = 1,   /* [p+1,B,d]  memories history                     (mac_cell.py:550,473) */
  MACX_SEG_INFOS = 2,      /* [p,B,d]    retrieved information per step       (mac_cell.py:474)     */
  MACX_SEG_ATT_QUESTION = 3,/* [p,B,S]   attentions["question"]               (mac_cell.py:176)     */
  MACX_SEG_ATT_KB = 4,     /* [p,B,N]    attentions["kb"]                     (mac_cell.py:268)     */
  MACX_SEG_ATT_SELF = 5,   /* [p,B,p]    attentions["self"], row i uses i+1   (mac_cell.py:329)     */
  MACX_SEG_ATT_GATE = 6,   /* [p,B,d|1]  attentions["gate"]                   (mac_cell.py:365)     */
  MACX_SEG_COUNT = 7
};

/* ---- sizing -------------------------------------------------------------------------------- */
/* floats the caller must provide in `saved` (kept from forward to backward) and `ws` (scratch).
 * keep_activations = 1 keeps the per-step [B,N,d] read-unit activations needed by backward. */
size_t macx_saved_floats(const macx_opts*, const macx_shapes*, int keep_activations);
size_t macx_ws_floats(const macx_opts*, const macx_shapes*, int for_backward);
/* offset (floats) and element count of a viewable segment of `saved`; returns MACX_OK or error */
int macx_saved_segment(const macx_opts*, const macx_shapes*, int keep_activations, int segment,
                       size_t* offset, size_t* count);
/* validates an option/shape combination exactly as macx_cell_begin would */
int macx_check(const macx_opts*, const macx_shapes*);

/* ---- the cell ------------------------------------------------------------------------------ */
/* Replaces MACCell.zero_state (mac_cell.py:539-592): initial control/memory, histories, memory
 * variational-dropout stream; additionally packs the weights for the MFMA kernels and, when the
 * control is not recurrent (controlFeedPrev off), computes all p control states up front
 * (mac_cell.py:442-451, :133-187). */
int macx_cell_begin(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*,
                    const macx_inputs*, float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                    int keep_activations, void* stream);

/* Replaces one MACCell.__call__ (mac_cell.py:420-480) for iteration `step`: control (if
 * recurrent), read, write, history append.  Must be called with step = 0..p-1 in order. */
int macx_cell_step(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*,
                   const macx_inputs*, float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                   int keep_activations, int step, void* stream);

/* begin + p steps in one call (the loop of model.py:453-458). */
int macx_cell_forward(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*,
                      const macx_inputs*, float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                      int keep_activations, void* stream);

/* Gradient of the whole p-step run (what tf.gradients builds for model.py:453-458).
 * d_memory / d_control: [B,d] gradients of the final state (either may be NULL = zero).
 * Requires `saved` from a forward run with keep_activations = 1, the same dropout struct, the same macx_opts (kernel family and
 * tuning table included) and the SAME PARAMETER VALUES: the weight matrices are read from the transposed / H2 packs the forward
 * call's pack launch left in `saved`, only biases and the attention vectors are read live from macx_params -- a run whose
 * parameters change between its forward and its backward call mixes old and new weights. */
int macx_cell_backward(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*,
                       const macx_inputs*, const float* saved, size_t saved_floats,
                       float* ws, size_t ws_floats,
                       const float* d_memory, const float* d_control,
                       const macx_param_grads*, const macx_input_grads*, void* stream);
/* The same in two parts for data-parallel overlap: phase 1 = everything except the read unit's deferred weight contractions
 * (dW2, dWx over all p*B*N rows, the S_b slab sums for dW1 and their bias sums), phase 2 = only those, on the buffers of a
 * phase-1 call; phase 0 = both (= macx_cell_backward).  After phase 1 every other parameter gradient and all input
 * gradients are final, so a host can all-reduce them on a side stream while phase 2 runs (mac-network_amd/dp.py). */
int macx_cell_backward_phase(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*,
                       const macx_inputs*, const float* saved, size_t saved_floats,
                       float* ws, size_t ws_floats,
                       const float* d_memory, const float* d_control,
                       const macx_param_grads*, const macx_input_grads*, int phase, void* stream);

/* ---- output unit + classifier (consumer of the final memory) ------------------ */
/* outputOp (model.py:512-528, --outQuestion) + classifier (model.py:547-576 -> ops.FCLayer
 * ops.py:349-359) for outClassifierDims = [hidden]:
 *   logits = dropout(act(dropout(concat([memory, vecQ Woq + boq])) W0 + b0)) W1 + b1
 * Variables: outputUnit/linearLayeroutQuestion, classifier/linearLayerfc_0, classifier/linearLayerfc_1. */
typedef struct macx_out_shapes {
  int32_t B, d, hidden, answers;  /* batch, memDim == ctrlDim, outClassifierDims[0], answerWordsNum */
  int32_t b0;                     /* global index of question 0 (dropout stream) */
} macx_out_shapes;
typedef struct macx_out_params {
  const float* outQuestion_W; const float* outQuestion_b;   /* [d,d], [d]              */
  const float* fc0_W; const float* fc0_b;                   /* [2d,hidden], [hidden]   */
  const float* fc1_W; const float* fc1_b;                   /* [hidden,answers], [answers] */
} macx_out_params;
typedef struct macx_out_grads {
  float* outQuestion_W; float* outQuestion_b; float* fc0_W; float* fc0_b; float* fc1_W; float* fc1_b;
} macx_out_grads;
size_t macx_output_saved_floats(const macx_out_shapes*);
size_t macx_output_ws_floats(const macx_out_shapes*);
/* act: resolved activation of "RELU" (config.relu); keep: outputDropout (1.0 in evaluation). */
int macx_output_forward(const macx_out_shapes*, int act, float keep, uint32_t seed, const macx_out_params*,
                        const float* memory, const float* vecQuestions, float* logits,
                        float* saved, size_t saved_floats, void* stream);
int macx_output_backward(const macx_out_shapes*, int act, float keep, uint32_t seed, const macx_out_params*,
                         const float* memory, const float* vecQuestions, const float* saved, size_t saved_floats,
                         float* ws, size_t ws_floats, const float* d_logits, const macx_out_grads*,
                         float* d_memory, float* d_vecQuestions, void* stream);

/* ---- stem CNN (producer of the knowledge base) -------------------------------- */
/* MACnet.stem (model.py:165-204) -> ops.CNNLayer / ops.cnn (ops.py:380-438) for the default
 * stemNumLayers = 2, stemKernelSize = 3, stride 1, no batch norm:
 *   KB = act(conv3x3_SAME(dropout(act(conv3x3_SAME(dropout(images), K0) + b0)), K1) + b1)
 * images [B, H*W, Cin] NHWC (config.imageDims = 14 x 14 x 1024), KB [B, H*W, Cout] (model.py:202).
 * Variables: stem/cnnLayercnn_{0,1}/kernels/kernel [3,3,in,out] (HWIO), .../biases/bias [out].
 * Cin, Cmid, Cout multiples of 128.  No gradient is returned for `images` (pre-extracted features). */
typedef struct macx_stem_shapes { int32_t B, H, W, Cin, Cmid, Cout, b0; } macx_stem_shapes;
typedef struct macx_stem_params { const float* kernel0; const float* bias0; const float* kernel1; const float* bias1; } macx_stem_params;
typedef struct macx_stem_grads { float* kernel0; float* bias0; float* kernel1; float* bias1; } macx_stem_grads;
size_t macx_stem_saved_floats(const macx_stem_shapes*);
size_t macx_stem_ws_floats(const macx_stem_shapes*);
int macx_stem_forward(const macx_stem_shapes*, int act, float keep, uint32_t seed, const macx_stem_params*,
                      const float* images, float* kb, float* saved, size_t saved_floats, void* stream);
int macx_stem_backward(const macx_stem_shapes*, int act, float keep, uint32_t seed, const macx_stem_params*,
                       const float* kb, const float* saved, size_t saved_floats, float* ws, size_t ws_floats,
                       const float* d_kb, const macx_stem_grads*, void* stream);

/* Feed-dict image layout (model.py:67-68): the h5 features are [B, C, H, W] (extract_features.py) and the graph
 * transposes them to NHWC before the stem.  nhwc[b][hw][c] = nchw[b][c][hw]. */
int macx_images_to_nhwc(const float* nchw, int B, int C, int HW, float* nhwc, void* stream);

/* ---- question encoder (producer of vecQuestions / questionCntxWords) ---------- */
/* qEmbeddingsOp (model.py:208-221) + encoder (model.py:255-307) for encType = LSTM, encBi,
 * encNumLayers = 1: embedding lookup (index 0 = zero pad row), dropout(encInputDropout), a
 * bidirectional BasicLSTMCell(h = encDim/2) under tf.nn.bidirectional_dynamic_rnn semantics,
 * questionCntxWords = concat(fw, bw) [B,S,2h], vecQuestions = dropout(concat(final h), qDropout).
 * Variables: qEmbeddings/emb [V,E]; encoder/birnnLayer/bidirectional_rnn/{fw,bw}/basic_lstm_cell/
 * {kernel [E+h,4h], bias [4h]} (gate order i, j, f, o; forget bias 1).  h % 128 == 0. */
typedef struct macx_enc_shapes { int32_t B, S, V, E, h, b0; } macx_enc_shapes;   /* V rows in emb (ids 1..V) */
typedef struct macx_enc_params { const float* emb; const float* fw_kernel; const float* fw_bias; const float* bw_kernel; const float* bw_bias; } macx_enc_params;
typedef struct macx_enc_grads { float* emb; float* fw_kernel; float* fw_bias; float* bw_kernel; float* bw_bias; } macx_enc_grads;
size_t macx_encoder_saved_floats(const macx_enc_shapes*);
size_t macx_encoder_ws_floats(const macx_enc_shapes*);
int macx_encoder_forward(const macx_enc_shapes*, float keep_input, float keep_question, uint32_t seed, const macx_enc_params*,
                         const int32_t* questions /*[B,S]*/, const int32_t* lengths /*[B]*/, float* words /*[B,S,2h]*/,
                         float* vecQuestions /*[B,2h]*/, float* saved, size_t saved_floats, void* stream);
int macx_encoder_backward(const macx_enc_shapes*, float keep_input, float keep_question, uint32_t seed, const macx_enc_params*,
                          const int32_t* questions, const int32_t* lengths, const float* saved, size_t saved_floats,
                          float* ws, size_t ws_floats, const float* d_words, const float* d_vecQuestions,
                          const macx_enc_grads*, void* stream);

/* ---- optimizer step ---------------------------------------------------------- */
/* addTrainingOp (model.py:639-669) over ONE flat fp32 buffer of n elements:
 *   norm = ||g||_2 ; g *= clip / max(norm, clip)      tf.clip_by_global_norm, clip_norm <= 0 disables
 *   Adam (tf.train.AdamOptimizer): m, v, p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
 *   ema -= (1 - decay) * (ema - p)                    tf.train.ExponentialMovingAverage, decay < 0 disables
 * `step` is 1-based; `ws` >= 1024 floats; `norm_out` (device, may be NULL) receives the pre-clip norm. */
int macx_adam_ema_step(size_t n, float* params, const float* grads, float* m, float* v, float* ema,
                       float lr, float beta1, float beta2, float eps, int step, float clip_norm, float ema_decay,
                       float* ws, float* norm_out, void* stream);

/* ---- unit-level entry points (the ops.py primitives; used by the parity tests) -------------- */
/* out[r, :] = act(concat(x1[r], x2[r]) @ W + b + bias_const)     ops.linear (ops.py:298-333)
 * on fp32 MFMA; `W_packed` from macx_pack_weight(W, k1 + k2, n_out, 0); k1, k2, n_out % 16 == 0. */
int macx_linear(const float* x1, int k1, const float* x2, int k2, int rows,
                const float* W_packed, const float* b, float bias_const, int n_out, int act,
                float* out, void* stream);
/* Re-lays a [K, n_out] weight (flags bit 0 = 0) or the transpose of a [n_out, K] weight (bit 0 = 1) into the operand
 * order a kernel reads; flags bits 1-2 select the format:
 *   MACX_PACK_F32MFMA (0)  out[Q][g][j][e] = W[16Q + 4g + e][j]            fp32, K*n_out floats   (macx_linear, native GEMM)
 *   MACX_PACK_BF16X3  (1)  out[K/32][3][n_out][32] bf16: the exact split W = W1 + W2 + W3 of every element into three
 *                          bf16 pieces (round-to-nearest residual chain), K*n_out*3/2 floats      (split GEMM)
 *   MACX_PACK_KMAJOR  (2)  out[K/32][n_out][32] fp32 k-major tiles, K*n_out floats
 * K % 16 == 0 (K % 32 for formats 1, 2), n_out % 16 == 0. */
#define MACX_PACK_TRANSPOSE 1
#define MACX_PACK_F32MFMA (0 << 1)
#define MACX_PACK_BF16X3 (1 << 1)
#define MACX_PACK_KMAJOR (2 << 1)
int macx_pack_weight(const float* W, int K, int n_out, int flags, float* out, void* stream);
/* Which kernel family runs the knowledge-base GEMMs (projX, memKbProj, memKbProj_2, their backward-data products, the
 * weight-gradient contractions and the stem's implicit-GEMM convolutions).  gfx950 issues f32-input MFMA at 1/16 of the
 * 16-bit rate and has no TF32 form, so the large contractions run on the 16-bit matrix pipe at fp32-class accuracy:
 *   MACX_GEMM_H2 (default)  every [B,N,d] activation is stored ONCE, by its producer, as x 2^e = hi + lo (two fp16 planes,
 *                           |error| <= 2^-24, one int8 exponent per row and 128 columns); a product is 3 MFMA terms on
 *                           v_mfma_f32_16x16x32_f16 with fp32 accumulation (see "the H2 tensor format" below)
 *   MACX_GEMM_SPLIT         every fp32 operand is split exactly into three bf16 pieces while it is staged, 6 terms on
 *                           v_mfma_f32_16x16x32_bf16
 *   MACX_GEMM_NATIVE        v_mfma_f32_16x16x4_f32 (bit-equal to an fmaf chain)
 * Measured error against fp64 of all three: tests/test_gpu_h2.py, tests/test_gpu_units.py.
 * This call sets / queries (mode < 0) the PROCESS DEFAULT and returns it; a run selects its own family with
 * macx_opts.gemm_family, which is applied per call on the calling thread.  Packed-weight formats and buffer sizes differ
 * between families: use one family for the sizing, forward and backward calls of a run. */
#define MACX_GEMM_NATIVE 0
#define MACX_GEMM_SPLIT 1
#define MACX_GEMM_H2 2
int macx_gemm_mode(int mode);

/* ---- the H2 tensor format (mode MACX_GEMM_H2, the default; mac-network_amd/csrc/macx_h2.hip.h) ---------------------
 * An fp32 [rows, cols] tensor (cols % 128 == 0) held as two fp16 planes in slot order + one int8 exponent per (row, 128
 * columns): x * 2^e = hi + lo to 2^-24.  Inside the cell every [B,N,d] activation lives in HBM in this format; these
 * entry points expose the conversions and one GEMM on it so that the format and its numerics are testable in isolation.
 *   macx_h2_floats      floats a caller-owned buffer needs for one H2 tensor (0 if cols % 128)
 *   macx_h2_from_f32    src [B][N][C] fp32 row-major -> h2
 *   macx_h2_to_f32      h2 -> out [rows][C] fp32 row-major
 *   macx_h2_pack_weight W [K, n_out] (or its transpose, from [n_out, K]) -> H2 weight planes + exponent; out >= K*n_out + 16 floats
 *   macx_h2_gemm_planes hO = act(hA @ Wh + bias) on operands already in the format (what the cell launches per step; timed by
 *                       bench.py's roofline probe)
 *   macx_h2_gemm        out[B*N, n_out] = act(A[B*N, K] @ W[K, n_out] + bias): from_f32 + pack + gemm_planes + to_f32 (three fp16
 *                       MFMA terms per product, fp32 accumulate); ws >= h2_floats(B*N,K) + h2_floats(B*N,n_out) + K*n_out + 16 */
size_t macx_h2_floats(size_t rows, size_t cols);
int macx_h2_from_f32(const float* src, int B, int N, int C, float* h2, void* stream);
int macx_h2_to_f32(const float* h2, int rows, int C, float* out, void* stream);
int macx_h2_pack_weight(const float* W, int K, int n_out, int transpose, float* out, void* stream);
int macx_h2_gemm_planes(const float* hA, int B, int N, int K, const float* Wh, int n_out, const float* bias, int act, float* hO,
                        void* stream);
int macx_h2_gemm(const float* A, int B, int N, int K, const float* W, int n_out, const float* bias, int act, float* out,
                 float* ws, size_t ws_floats, void* stream);
/* bench / profiling hook (mode MACX_GEMM_H2, d <= 512, N >= 16): the kernel that runs the read unit's three knowledge-base
 * products of one step -- dropout(KB) -> X -> H1 -> I2 -> attention logits (mac_cell.py:230-266, ops.py:668-725;
 * mac-network_amd/csrc/macx_chain_h2.hip.h) -- re-launched `reps` times on `stream` between two HIP events; *ms_out = average
 * milliseconds per launch.  `saved` comes from macx_cell_begin + macx_cell_step(.., step) with keep = 1; launch r writes the
 * buffers of step (step + r) % p, so the run must not be differentiated afterwards.  MACX_EUNSUPPORTED when this shape /
 * family does not run on that kernel. */
int macx_read_chain_time(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*, const macx_inputs*,
                         float* saved, size_t saved_floats, int step, int reps, float* ms_out, void* stream);
/* ... and IN A RUNNING FORWARD PASS: one macx_cell_forward (keep = 1) on `stream`, each of its p chain launches issued with a start
 * and a stop HIP event that receive the kernel's own dispatch timestamps (the step's [B,d] linear in front of a launch, the
 * attention kernel behind it, as in a training step), behind three untimed passes so that the chip is as busy as inside a training
 * loop; *ms_out = average milliseconds per launch of the timed pass.  Synchronises `stream`.
 * bench.py's roofline.kernel_ms. */
int macx_cell_forward_chain_time(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*, const macx_inputs*,
                                 float* saved, size_t saved_floats, float* ws, size_t ws_floats, float* ms_out, void* stream);
/* One of the read unit's kept [B*N, d] activations of step `step`, as fp32 row-major, from the `saved` buffer of a forward pass with
 * keep = 1: which = 0 dropout(KB) (ops.py:678), 1 X = dropout(KB) Wx + bx (ops.py:688), 2 H1 (ops.py:718, mac_cell.py:237), 3 I2
 * (ops.py:326) -- whatever format the kernel family keeps them in (H2 planes in the default family).  Inspection and tests: the chain
 * kernel's intermediate products are checked against fp64 through this.  out: [B*N][d] floats, 16-byte aligned. */
int macx_saved_activation(const macx_opts*, const macx_shapes*, int which, int step, const float* saved, size_t saved_floats,
                          float* out, void* stream);
/* The control unit's question projections on their own (mac_cell.py:442-448; SURVEY 8b's `ctrl_inputs` unit):
 *   fwd:  ctrl_t[B,d] = controlInputAct(vecQuestions Wq + bq);  ctrl_inputs[p,B,d]: step i = ctrl_t WqU_i + bqU_i (one matrix per
 *         step with controlInputUnshared, else the shared one) -- the two launches macx_cell_begin issues for them;
 *   bwd:  d_ctrl_inputs[p,B,d] -> d_vecQuestions[B,d] and the non-NULL ones of GP->qInput_W / qInput_b / qInputU_W / qInputU_b.
 * Only the qInput / qInputU fields of macx_params / macx_param_grads are touched.  ws >= macx_ctrl_inputs_ws_floats() floats
 * (packed weights and [B,d] temporaries; nothing survives from fwd to bwd in it -- pass ctrl_t back in). */
size_t macx_ctrl_inputs_ws_floats(const macx_opts*, const macx_shapes*);
int macx_ctrl_inputs_fwd(const macx_opts*, const macx_shapes*, const macx_params*, const float* vecQuestions, float* ctrl_t,
                         float* ctrl_inputs, float* ws, size_t ws_floats, void* stream);
int macx_ctrl_inputs_bwd(const macx_opts*, const macx_shapes*, const macx_params*, const float* vecQuestions, const float* ctrl_t,
                         const float* d_ctrl_inputs, const macx_param_grads*, float* d_vecQuestions, float* ws, size_t ws_floats,
                         void* stream);
/* X = dropout(KB) @ Wx + bx: the projX half of ops.mul (ops.py:678,688) on the knowledge-base GEMM.
 * `W_packed` from macx_pack_weight(Wx, d, d, macx_gemm_mode(-1) ? MACX_PACK_BF16X3 : MACX_PACK_F32MFMA);
 * `drop_ws` >= B*N*d + B*N*d/32 floats of scratch for the dropped KB and its keep bits (may be NULL when keep_read == 1). */
int macx_kb_project(const macx_shapes*, const macx_dropout*, int step, const float* kb,
                    const float* W_packed, const float* b, float* out, float* drop_ws, void* stream);
/* softmax(expMask(logits)) + att2Smry over the question words for one step
 * (mac_cell.py:155-181; ops.py:114-150, 243-247).  cc: continuous control [B,d]. */
int macx_control_attend(const macx_shapes*, const float* cc, const float* words, const int32_t* lengths,
                        const float* w, const float* b, float* att, float* control, void* stream);
/* its backward: d_control[B,d] -> d_cc[B,d], d_words[B,S,d] (written), d_w[d], d_b[1]; `att` is what the forward call
 * returned; ws >= macx_control_attend_bwd_ws_floats(shapes) floats.  d % 64 == 0. */
size_t macx_control_attend_bwd_ws_floats(const macx_shapes*);
int macx_control_attend_bwd(const macx_shapes*, const float* d_control, const float* cc, const float* att, const float* words,
                            const float* w, float* ws, size_t ws_floats, float* d_cc, float* d_words, float* d_w, float* d_b,
                            void* stream);
/* Materialises the 0/1 keep mask of a dropout site for n elements starting at flat index
 * `first` (test hook for the stateless stream; site numbers in macx_common.hip.h). */
int macx_dropout_mask(uint32_t seed, uint32_t site, uint32_t step, float keep, uint32_t first,
                      size_t n, float* out, void* stream);
/* ... under a run's mask word (macx_dropout.mask_word: device pointer or NULL) */
int macx_dropout_mask_w(uint32_t seed, uint32_t site, uint32_t step, float keep, uint32_t first,
                        size_t n, const uint32_t* mask_word, float* out, void* stream);
/* weight-gradient contraction out[k][j] = sum_m A[m][k] G[m][j]  (fixed-order split reduction).
 * `ws` >= nsplit*Kd*Jd floats where nsplit = macx_wgrad_splits(M, Kd, Jd). */
int macx_wgrad_splits(int M, int Kd, int Jd);
int macx_wgrad(const float* A, int lda, const float* G, int ldg, int M, int Kd, int Jd,
               float* out, float* ws, void* stream);

/* ---- answer loss and prediction (addAnswerLossOp model.py:593-599, addPredOp model.py:603-612) ----
 * loss_rows[b] = -log softmax(logits[b])[answers[b]]  (tf.nn.sparse_softmax_cross_entropy_with_logits; the model's loss is
 * their mean), pred[b] = argmax_c logits[b][c] (first maximum, tf.argmax), dlogits (may be NULL) = (softmax - onehot) *
 * grad_scale -- grad_scale = 1/B is the gradient of the mean loss.  logits [B][A] fp32 contiguous, answers / pred int32. */
int macx_answer_loss(const float* logits, const int32_t* answers, int B, int A, float* loss_rows, int32_t* pred,
                     float* dlogits, float grad_scale, void* stream);

/* ---- the knowledge-base attention unit on its own (mac_cell.py:266-272: inter2att's softmax + att2Smry) ------------
 * The fused cell's own kernels behind a per-unit contract, so that the unit is testable in isolation:
 *   macx_kb_attend_fwd   att[B,N] = softmax_n(logits[B,N] + bias[0]);  info[B,d] = sum_n att KB        (ops.py:140-150)
 *   macx_kb_attend_bwd   da = dinfo . KB;  dlogits = att (da - sum_n att da);  dkb (= | +=) att (x) dinfo (dkb may be NULL);
 *                        ws >= macx_kb_attend_bwd_ws_floats(B, N, d) floats
 * N <= 1024, d % 128 == 0. */
int macx_kb_attend_fwd(int B, int N, int d, const float* logits, const float* bias, const float* kb, float* att, float* info,
                       void* stream);
size_t macx_kb_attend_bwd_ws_floats(int B, int N, int d);
int macx_kb_attend_bwd(int B, int N, int d, const float* att, const float* kb, const float* dinfo, float* dlogits, float* dkb,
                       int accumulate, float* ws, size_t ws_floats, void* stream);

/* ---- the read unit and the write unit as wholes (macx_<unit>_{fwd,bwd}) ----------------------------------
 * ONE unit of the fused cell on caller-owned buffers -- the cell's own step code restricted to that unit, so per-unit
 * parity against mac_cell.py:209-277 (read) and :305-375 (write) is testable without running a cell.
 *   shapes->p must be 1: it is the unit of step 0 (dropout streams are keyed by (seed, site, step 0)); S is ignored.
 *   saved >= macx_saved_floats(opts, shapes, 1) floats, written by *_fwd and read by the matching *_bwd;
 *   ws >= macx_workspace_bytes(opts, shapes, 1) bytes (= 4 * macx_ws_floats).  Only the unit's fields of macx_params /
 *   macx_param_grads are read / written (read: projX, projY, memKbProj, memKbProj2, kbLogits; write: newMemory, gate).
 *   read:  info[B,d], att[B,N] = read(knowledgeBase[B,N,d], memory[B,d], control[B,d]); memory / read dropout per macx_dropout
 *          bwd: d_info -> d_knowledgeBase[B,N,d], d_memory, d_control and the unit's parameter gradients
 *   write: new_memory[B,d] = write(memory, info, control); the write dropout (mac_cell.py:461-463) is applied to `info`
 *          bwd: d_new_memory -> d_memory, d_info, d_control (the gate's; zeros without --writeGate) + parameter gradients
 *          opts->write_self_att needs the histories of a running cell: MACX_EUNSUPPORTED here (use macx_cell_step). */
size_t macx_workspace_bytes(const macx_opts*, const macx_shapes*, int for_backward);
int macx_read_fwd(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*, const float* knowledgeBase,
                  const float* memory, const float* control, float* saved, size_t saved_floats, float* info, float* att,
                  void* stream);
int macx_read_bwd(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*, const float* knowledgeBase,
                  const float* saved, size_t saved_floats, float* ws, size_t ws_floats, const float* d_info,
                  const macx_param_grads*, float* d_knowledgeBase, float* d_memory, float* d_control, void* stream);
int macx_write_fwd(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*, const float* memory,
                   const float* info, const float* control, float* saved, size_t saved_floats, float* new_memory, void* stream);
int macx_write_bwd(const macx_opts*, const macx_shapes*, const macx_dropout*, const macx_params*, const float* saved,
                   size_t saved_floats, float* ws, size_t ws_floats, const float* d_new_memory, const macx_param_grads*,
                   float* d_memory, float* d_info, float* d_control, void* stream);

/* ---- embedding lookup on its own (model.py:207-219 qEmbeddingsOp + the input dropout of ops.py:812 / :880) --------------
 *   macx_embed_lookup      x[r][0..E) = dropout(table[ids[r]]), table row 0 = zeros (padding), row i = emb[i - 1]; columns
 *                          [E, ld) of x are written as zeros.  Dropout site SITE_ENC_INPUT on the stateless stream, element
 *                          index (first_row + r) * E + c (first_row = b0 * S for a data-parallel shard); keep = 1: none.
 *   macx_embed_lookup_bwd  d_emb[V][E] (written) = sum over the rows that looked a word up of dropout'(dx[r]); one workgroup
 *                          per vocabulary row, fixed order, no atomics.
 * The generic question encoder (mac-network_amd/encoder.py GenericQuestionEncoder) is built from these, macx_linear / macx_wgrad
 * and the macx_op_* kernels. */
int macx_embed_lookup(const int32_t* ids, const float* emb, int rows, int E, int ld, float keep, uint32_t seed,
                      uint32_t first_row, float* x, void* stream);
int macx_embed_lookup_bwd(const int32_t* ids, const float* dx, int rows, int E, int ld, int V, float keep, uint32_t seed,
                          uint32_t first_row, float* d_emb, void* stream);

/* ---- the ops.py primitives as single kernels (mac-network_amd/csrc/macx_ops.hip.h) -------------------------------
 * The building blocks of the GENERIC option path (mac-network_amd/generic.py): every legal option combination the fused
 * cell kernels above answer with MACX_EUNSUPPORTED runs as one kernel per reference op -- these, macx_linear / macx_h2_gemm
 * and macx_wgrad -- chained by the host exactly as mac_cell.py chains ops.py.  fp32, contiguous, device pointers.
 *   macx_op_act      out = act(x); act = MACX_ACT_* or MACX_OP_PRELU (relu(x) - alpha[c] relu(-x), c = index % inner; ops.py:171-173)
 *   macx_op_act_bwd  dx = dy act'(x); PRELU also writes dalpha_elem = dy min(x, 0) for the caller to reduce over rows
 *   macx_op_binary   out = scale (a + b) or scale (a * b); a, out hold n floats viewed [.., mid, inner]; b by `bmode`:
 *                    SAME [n] | MID [n / (mid inner)][inner], broadcast over the middle axis (ops.mul's extendY, ops.py:693-695) |
 *                    CHANNEL [inner] | ROW [n / inner], one scalar per row (ops.att2Smry, ops.py:150)
 *   macx_op_reduce   MID: x [outer][mid][inner] -> out [outer][inner];  LAST: x [outer][inner] -> out [outer];
 *                    ROWS: x [outer][inner] -> out [inner], ws >= 64 inner floats.  Fixed summation order.
 *   macx_op_softmax  softmax over the last axis of [rows][n]; with `lengths`, columns >= lengths[row / rows_per_len] are
 *                    masked to -inf first (ops.expMask, ops.py:243-247; mac_cell.py:176)
 *   macx_op_softmax_bwd  dx = a (da - sum a da)
 *   macx_op_dropout  out = x / keep * mask(seed, site, step, first + index): tf.nn.dropout on the stateless stream of
 *                    macx_dropout_mask (ops.py:312, mac_cell.py:217,463); its own backward (apply it to dy) */
#define MACX_OP_PRELU 16
#define MACX_OP_RSQRT_EPS 17   /* out = 1 / sqrt(x + alpha[0]): batch-norm normaliser (mac_cell.py:370-373); no dalpha */
enum { MACX_OP_ADD = 0, MACX_OP_MUL = 1 };
enum { MACX_OP_B_SAME = 0, MACX_OP_B_MID = 1, MACX_OP_B_CHANNEL = 2, MACX_OP_B_ROW = 3 };
enum { MACX_OP_R_MID = 0, MACX_OP_R_LAST = 1, MACX_OP_R_ROWS = 2 };
int macx_op_act(int act, const float* x, const float* alpha, size_t n, int inner, float* out, void* stream);
int macx_op_act_bwd(int act, const float* x, const float* alpha, const float* dy, size_t n, int inner, float* dx,
                    float* dalpha_elem, void* stream);
int macx_op_binary(int op, int bmode, const float* a, const float* b, size_t n, int mid, int inner, float scale, float* out,
                   void* stream);
int macx_op_reduce(int mode, const float* x, size_t outer, int mid, int inner, float* out, float* ws, void* stream);
int macx_op_softmax(const float* x, const int32_t* lengths, int rows_per_len, size_t rows, int n, float* out, void* stream);
int macx_op_softmax_bwd(const float* a, const float* da, size_t rows, int n, float* dx, void* stream);
int macx_op_dropout(const float* x, size_t n, uint32_t seed, uint32_t site, uint32_t step, float keep, uint32_t first,
                    float* out, void* stream);
/* ... under a run's mask word (macx_dropout.mask_word: device pointer or NULL) */
int macx_op_dropout_w(const float* x, size_t n, uint32_t seed, uint32_t site, uint32_t step, float keep, uint32_t first,
                      const uint32_t* mask_word, float* out, void* stream);

/* (ABI <= 4 had a `macx_debug_set` entry point here: fifteen process-global integers.  ABI 5 removed it -- the live A/B hooks travel per
 * call in macx_opts.tune, the variants that were measured slower in rounds 4-5 (side-queue modes, pair launches, the dual-A
 * contraction, 8-wave linears, the 32 x 32 x 16 K loop, write-through stores) and the kernels only they instantiated are gone.
 * What is left of process-wide state is macx_gemm_mode's default kernel FAMILY for the entry points that carry no macx_opts;
 * no product module sets it: tests/test_host.py::test_product_code_never_touches_the_debug_knobs.) */

const char* macx_strerror(int code);
int macx_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MACX_H */
