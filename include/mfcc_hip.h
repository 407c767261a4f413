/*
 * mfcc_hip.h -- C ABI of the MI355X-native MFCC hot path (libmfcc_hip.so).
 *
 * Drop-in boundary for the per-frame MFCC math of lambdaconcept/mfcc's `mfcc/core`
 * (pre-emphasis -> framing -> Hamming -> FFT -> |.|^2 -> mel -> log2 -> DCT-II -> keep n_cep).
 * The reference has no software operator API for this path: it sits behind
 *   (1) the RTL stream interface  MFCC.sink / MFCC.source / MFCC.reset   mfcc/core/mfcc.py:28-30
 *   (2) the host driver's C calls  mfcc_open / mfcc_convert / mfcc_close  software/main.c:36,100,53
 *       over the transport           ft601_write / ft601_read             software/ft601.h:55-56
 * This header is the batch equivalent of (2): one call = whole utterance(s) instead of a
 * USB ping-pong per frame.  Plain pointers and sizes only; no C++ / torch types.
 *
 * Conventions kept from the reference:
 *   - 0 = success, negative error codes in the ft601_error range      software/ft601.h:25-32
 *   - caller owns every in/out buffer; the handle owns device tables/streams   main.c:109,40
 *   - output layout [frame][n_cep] row-major, i.e. the `.mfcc` file layout     main.c:162-165
 *   - nothing is printed by the library (the reference printf's; a log hook exists there,
 *     ft601.h:43-51 -- here errors are returned and described by mfcc_hip_strerror)
 *   - a handle is not thread-safe; distinct handles are independent            ft601.c:185,197
 *
 * Two numeric contracts:
 *   float : notebook/MFCC.ipynb (float64 NumPy) evaluated in fp32 on the GPU, <= 1e-4 rel-err
 *   fixed : the nMigen RTL arithmetic of mfcc/core + mfcc/misc/fft.py, bit-exact int16
 */
#ifndef MFCC_HIP_H
#define MFCC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFCC_HIP_ABI_VERSION 2

/* error codes: same numbering family as `enum ft601_error` (software/ft601.h:25-32) */
enum mfcc_hip_error {
    MFCC_HIP_SUCCESS             = 0,
    MFCC_HIP_ERROR_INVALID_PARAM = -101,
    MFCC_HIP_ERROR_NOT_FOUND     = -102,   /* no usable HIP device                     */
    MFCC_HIP_ERROR_NO_MEM        = -103,
    MFCC_HIP_ERROR_BUSY          = -104,
    MFCC_HIP_ERROR_UNSUPPORTED   = -105,   /* parameter combination has no kernel      */
    MFCC_HIP_ERROR_BUFFER_SMALL  = -106,   /* caller's output capacity is too small    */
    MFCC_HIP_ERROR_IO            = -107,   /* file open/read/write (wav -> .mfcc)      */
    MFCC_HIP_ERROR_OTHER         = -200    /* HIP runtime error, see mfcc_hip_last_hip_error */
};

/* framing of the tail of a stream */
enum mfcc_hip_pad_mode {
    /* notebook/MFCC.ipynb cell 9: frames = int((n - nfft) / hop) + 1, tail samples dropped */
    MFCC_HIP_PAD_NOTEBOOK = 0,
    /* host driver + RTL bench: zeros are fed after EOF until the frame holding the last
     * sample is out: frames = (n - nfft) / hop + 2 (1 if n < nfft)   software/main.c:95,134-144 */
    MFCC_HIP_PAD_STREAM = 1
};

/* which float kernel to run (the results agree to fp32 rounding) */
enum mfcc_hip_float_impl {
    MFCC_HIP_IMPL_AUTO = 0,     /* fastest kernel that supports the parameters          */
    MFCC_HIP_IMPL_GENERIC = 1,  /* one frame per wave, any supported nfft / n_mel       */
    MFCC_HIP_IMPL_FUSED512 = 2  /* 512/170/32 specialised kernel                        */
};

/*
 * Parameters = the constructor arguments of `MFCC(width=16, nfft, samplerate, nfilters,
 * nceptrums)` (mfcc/core/mfcc.py:20-21) plus the host driver's constants
 * (`NFFT 512, STEPSIZE 170, NCEPSTRUMS, SAMPLERATE 16000`, software/main.c:11-14).
 */
typedef struct mfcc_hip_params {
    uint32_t struct_size;  /* = sizeof(mfcc_hip_params); set by mfcc_hip_default_params   */
    int32_t  nfft;         /* 512.  float: 256/512/1024; fixed: power of two 64..1024     */
    int32_t  hop;          /* 170.  0 -> nfft / 3 (mfcc/core/mfcc.py:43)                  */
    int32_t  n_mel;        /* 32.   float: 1..64; fixed: 4*n_mel must be a power of two   */
    int32_t  n_cep;        /* 13.   1..n_mel; `Discard(first=0, count)` misc/discard.py   */
    int32_t  sample_rate;  /* 16000                                                       */
    int32_t  pad_mode;     /* enum mfcc_hip_pad_mode                                      */
    float    power_scale;  /* float path: P = |X / power_scale|^2.  The notebook hard-codes
                              512 (cell 22); 0 -> nfft                                    */
    float    lifter;       /* float path: sinusoidal lifter L (MFCC.ipynb cell 43,
                              software/lift.py:12); 0 = off                               */
    int32_t  device;       /* HIP device ordinal; -1 = the current device                 */
    int32_t  float_impl;   /* enum mfcc_hip_float_impl                                    */
    int32_t  reserved[5];  /* must be zero                                                */
} mfcc_hip_params;

typedef struct mfcc_hip_handle mfcc_hip_handle;

/* ---- lifetime: replaces mfcc_open / mfcc_close (software/main.c:36-56) ------------- */

int  mfcc_hip_abi_version(void);
/* fills *p with nfft 512, hop 170, n_mel 32, n_cep 13, 16 kHz, NOTEBOOK, scale 512 */
int  mfcc_hip_default_params(mfcc_hip_params *p);
/* validates, builds the constant tables on the host, uploads them, creates a stream.
 * Fails with MFCC_HIP_ERROR_NOT_FOUND when no GPU is present: there is no CPU fallback. */
int  mfcc_hip_create(const mfcc_hip_params *p, mfcc_hip_handle **out);
/* Lifetime rule: a handle that still has streaming sessions (mfcc_hip_stream_create below) is only MARKED
 * by mfcc_hip_destroy -- its stream, tables and scratch stay valid for those sessions, no new session can be
 * opened on it and no other call may be made with it -- and is freed by the mfcc_hip_stream_destroy of its
 * last session.  Without live sessions it is freed at once.  Either order of the destroy calls is safe. */
void mfcc_hip_destroy(mfcc_hip_handle *h);
/* run on a caller-provided hipStream_t (e.g. torch's current stream).  NULL means the HIP
 * null (default) stream -- which is what torch uses unless told otherwise -- NOT "none". */
int  mfcc_hip_set_stream(mfcc_hip_handle *h, void *hip_stream);
/* go back to the handle's own (non-blocking) stream, the state after mfcc_hip_create */
int  mfcc_hip_use_own_stream(mfcc_hip_handle *h);
int  mfcc_hip_synchronize(mfcc_hip_handle *h);

/* ---- host-only helpers (work without a GPU) ------------------------------------------ */

/* frame count for a stream of n_samples under p->pad_mode (`nframes`, main.c:95) */
int  mfcc_hip_num_frames(const mfcc_hip_params *p, size_t n_samples, size_t *n_frames);
const char *mfcc_hip_strerror(int err);
/* hipError_t of the last failing runtime call on this handle (0 if none); with h == NULL:
 * of the last failing mfcc_hip_create on this thread */
int  mfcc_hip_last_hip_error(const mfcc_hip_handle *h);

/* The constant tables the kernels use, as built on the host (no GPU needed) -- lets the
 * CPU test-suite check the table builders against the oracle.  `which`: */
enum mfcc_hip_table {
    MFCC_HIP_TABLE_WINDOW_F32      = 0,  /* float[nfft]            periodic Hamming       */
    MFCC_HIP_TABLE_MEL_POINTS_I32  = 1,  /* int32[n_mel + 2]       filter points          */
    MFCC_HIP_TABLE_MEL_DENSE_F32   = 2,  /* float[n_mel][nfft/2+1] weights (unscaled)     */
    MFCC_HIP_TABLE_DCT_F32         = 3,  /* float[n_cep][n_mel]    ortho DCT-II (x lifter)*/
    MFCC_HIP_TABLE_FX_CURVE_I32    = 4,  /* int32[nfft]            RTL window curve       */
    MFCC_HIP_TABLE_FX_TWIDDLE_I32  = 5,  /* int32[nfft/2][2]       RTL twiddle ROM re,im  */
    MFCC_HIP_TABLE_FX_MEL_DENSE_U32 = 6  /* uint32[n_mel][nfft/2]  RTL filterbank weights
                                            (x 2^-30), closed form of the accumulators    */
};
/* writes up to `cap_bytes`; *n_bytes = size of the table */
int  mfcc_hip_get_table(const mfcc_hip_params *p, int which, void *buf, size_t cap_bytes,
                        size_t *n_bytes);

/* ---- the hot path: replaces the per-frame ft601_write / ft601_read loop of
 *      mfcc_convert (software/main.c:128-166) ------------------------------------------ */

/*
 * Host buffers.  pcm: [n_channels][n_samples_per_ch] int16 (each channel is an independent
 * stream: pre-emphasis history starts at 0, as after `mfcc_softreset`, main.c:21-34).
 * out: [n_channels][n_frames][n_cep].  out_capacity counts elements of out.
 * Synchronous: returns after the results are in `out`.
 */
int  mfcc_hip_process_i16(mfcc_hip_handle *h, const int16_t *pcm, size_t n_samples_per_ch,
                          size_t n_channels, float *out, size_t out_capacity, size_t *n_frames);
int  mfcc_hip_process_fixed_i16(mfcc_hip_handle *h, const int16_t *pcm, size_t n_samples_per_ch,
                                size_t n_channels, int16_t *out, size_t out_capacity,
                                size_t *n_frames);

/*
 * Ragged batch (host buffers): n_utterances utterances of different lengths in one launch -- the
 * batched form of the driver's directory walk (show_dir_content -> mfcc_convert per file,
 * software/main.c:206-247), every utterance an independent stream starting from reset.
 * pcm: the utterances back to back; utterance u is pcm[offsets[u] .. offsets[u + 1]), offsets has
 * n_utterances + 1 entries.  out: the frames of all utterances back to back, [sum frames][n_cep];
 * utterance u's frames are rows frame_offsets[u] .. frame_offsets[u + 1] (n_utterances + 1 entries,
 * written even when out is too small, so a caller can size out from a first call with capacity 0:
 * MFCC_HIP_ERROR_BUFFER_SMALL).  Results are bit-identical to one mfcc_hip_process_* call per
 * utterance.  Synchronous.
 */
int  mfcc_hip_process_ragged_i16(mfcc_hip_handle *h, const int16_t *pcm, const size_t *offsets,
                                 size_t n_utterances, float *out, size_t out_capacity,
                                 size_t *frame_offsets);
int  mfcc_hip_process_ragged_fixed_i16(mfcc_hip_handle *h, const int16_t *pcm, const size_t *offsets,
                                       size_t n_utterances, int16_t *out, size_t out_capacity,
                                       size_t *frame_offsets);

/* The same with the utterances and the result in HBM (d_pcm: all utterances back to back, d_out: dense
 * [sum frames][n_cep]); offsets / frame_offsets stay host arrays.  Asynchronous on the handle's stream
 * like the other *_dev entry points; frame_offsets is complete on return. */
int  mfcc_hip_process_ragged_i16_dev(mfcc_hip_handle *h, const void *d_pcm, const size_t *offsets,
                                     size_t n_utterances, void *d_out, size_t out_capacity,
                                     size_t *frame_offsets);
int  mfcc_hip_process_ragged_fixed_i16_dev(mfcc_hip_handle *h, const void *d_pcm, const size_t *offsets,
                                           size_t n_utterances, void *d_out, size_t out_capacity,
                                           size_t *frame_offsets);

/*
 * Device-resident buffers (HBM in, HBM out), asynchronous on the handle's stream.
 * d_pcm:  channel c starts at d_pcm + c * ch_stride_samples (int16 units).
 * halo:   0 or 1.  1 = the first sample of every channel is only pre-emphasis history
 *         (x[-1] of a shard cut out of a longer stream); frames start at sample 1 and
 *         n_samples_per_ch does not count it.  Used for frame-range sharding.
 * d_out:  [n_channels][n_frames][n_cep], float (float path) or int16 (fixed path).
 */
int  mfcc_hip_process_i16_dev(mfcc_hip_handle *h, const void *d_pcm, size_t n_samples_per_ch,
                              size_t ch_stride_samples, size_t n_channels, int halo,
                              void *d_out, size_t *n_frames);
int  mfcc_hip_process_fixed_i16_dev(mfcc_hip_handle *h, const void *d_pcm,
                                    size_t n_samples_per_ch, size_t ch_stride_samples,
                                    size_t n_channels, int halo, void *d_out, size_t *n_frames);

/*
 * Measurement helper: `iters` back-to-back launches of the float (fixed = 0) or fixed
 * (fixed = 1) kernel on device-resident buffers, bracketed by HIP events recorded on the
 * stream the kernel is launched on; *avg_ms = elapsed / iters.  Used by bench.py for the
 * `roofline.achieved` figure.
 */
int  mfcc_hip_time_dev(mfcc_hip_handle *h, int fixed, const void *d_pcm, size_t n_samples_per_ch,
                       size_t ch_stride_samples, size_t n_channels, void *d_out,
                       int warmup, int iters, float *avg_ms);

/* name of the kernel symbol process_*_dev launches for this handle (to match rocprofv3 rows) */
const char *mfcc_hip_kernel_name(const mfcc_hip_handle *h, int fixed);

/* ---- file-level convenience: mfcc_convert(sess, wav_in, mfcc_out)  software/main.c:100 --- */

/* 16-bit mono PCM WAV at p->sample_rate -> raw int16 LE `.mfcc` file [frame][n_cep]
 * (fixed = 1: RTL-exact values, what the FPGA would have written; fixed = 0: float
 * coefficients truncated to int16 like software/lift.py:39).  *n_frames_out may be NULL. */
int  mfcc_hip_convert_wav(mfcc_hip_handle *h, const char *wav_in, const char *mfcc_out,
                          int fixed, size_t *n_frames_out);

/* The same for n_files files at once -- the whole directory walk of show_dir_content (main.c:206-247)
 * as ONE ragged launch; every file is an independent stream, the files written are byte-identical to
 * n_files calls of mfcc_hip_convert_wav.  n_frames_each (n_files entries) may be NULL. */
int  mfcc_hip_convert_wavs(mfcc_hip_handle *h, const char *const *wav_in, const char *const *mfcc_out,
                           size_t n_files, int fixed, size_t *n_frames_each);

/* ---- online / streaming session: the core's real interface is a stream with state --------------
 * `MFCC.sink` (samples in), `MFCC.source` (coefficients out, one `first..last` burst per frame) and
 * `MFCC.reset` (mfcc/core/mfcc.py:28-30,116); the targets feed it chunk by chunk
 * (mfcc/targets/wav2mfcc.py:27-42: bit 31 of a word = soft reset; mic2mfcc.py:19-30: I2S samples
 * through a FIFO) and the receiver reads columns as they come (software/cepstrum.c:93-159).
 * A session carries exactly the core's cross-frame state on the device between calls: the one
 * pre-emphasis history sample (preemph.py:20-28) and the samples of the frame in progress (the ring
 * buffer of frame.py:65-153, at most nfft - 1 + hop of them).  Any chunking of a stream gives, frame
 * for frame, bit for bit, the result of the one-shot calls above (tests/test_gpu_parity.py).
 * Several sessions may share a handle; like the handle they are not thread-safe.                 */
typedef struct mfcc_hip_stream mfcc_hip_stream;

/* fixed = 0: float contract (out = float), fixed = 1: RTL contract (out = int16_t) */
int  mfcc_hip_stream_create(mfcc_hip_handle *h, int fixed, mfcc_hip_stream **out);
/* frees the session's device buffers; if its handle was already given to mfcc_hip_destroy and this was the
 * handle's last session, the handle is freed here (see the lifetime rule at mfcc_hip_destroy) */
void mfcc_hip_stream_destroy(mfcc_hip_stream *s);
/* `mfcc_softreset` (software/main.c:21-34): drop the pending samples, history back to 0 */
int  mfcc_hip_stream_reset(mfcc_hip_stream *s);
/* samples waiting for their frame to complete (0 <= pending < nfft between calls) */
size_t mfcc_hip_stream_pending(const mfcc_hip_stream *s);
/* upper bound of the frames a push of n samples can complete: (pending + n) / hop + 1 */
size_t mfcc_hip_stream_max_frames(const mfcc_hip_stream *s, size_t n);
/*
 * Feed n samples (host buffer); every frame they complete is computed and written to `out`
 * ([frames][n_cep], float or int16_t by the session's contract; out_capacity in elements).
 * *n_frames_out = frames written (may be 0).  MFCC_HIP_ERROR_BUFFER_SMALL leaves the session
 * untouched.  Synchronous.
 */
int  mfcc_hip_stream_push(mfcc_hip_stream *s, const int16_t *samples, size_t n, void *out,
                          size_t out_capacity, size_t *n_frames_out);
/*
 * End of the stream.  MFCC_HIP_PAD_STREAM: the host driver keeps feeding zeros until the frame
 * holding the last sample is out (main.c:134-144) -- one more, zero-padded frame is written.
 * MFCC_HIP_PAD_NOTEBOOK: the tail samples are dropped (notebook cell 9), nothing is written.
 * Either way the session is back in its reset state afterwards.
 */
int  mfcc_hip_stream_flush(mfcc_hip_stream *s, void *out, size_t out_capacity, size_t *n_frames_out);

/* ---- `.mfcc` -> `.lift` (software/lift.py:28-40): host only, no GPU --------------------------
 * reads raw int16 [frame][n_cep], multiplies column n by 1 + (L/2) sin(pi n / L) in double
 * (lift.py:12-26; L <= 0: unchanged) and writes `astype(np.int16)` of it: truncation toward zero,
 * low 16 bits for values beyond int16 (what NumPy does on x86-64).  *n_frames_out may be NULL. */
int  mfcc_hip_lift_file(const char *mfcc_in, const char *lift_out, int n_cep, double L,
                        size_t *n_frames_out);

/* ---- serial wire format of the FPGA's coefficient stream (host only, no GPU) ---------------
 * mfcc/misc/magic.py:9-41 (MagicInserter: 0xa55a in front of every frame's coefficients),
 * software/serial.c:13-14,89-122 (expect_magic: byte-wise resynchronisation, big endian),
 * software/cepstrum.c:15-71 (cepstrum_get_column: magic, then n_cep big-endian int16).       */

/* bytes mfcc_hip_serial_pack writes for n_frames frames: n_frames * 2 * (n_cep + 1) */
size_t mfcc_hip_serial_packed_size(size_t n_frames, int n_cep);

/* cep [n_frames][n_cep] int16 (what process_fixed_i16 returns) -> the byte stream the FPGA's
 * UART carries: per frame 0xa5 0x5a, then n_cep coefficients high byte first.             */
int  mfcc_hip_serial_pack(const int16_t *cep, size_t n_frames, int n_cep, uint8_t *out, size_t out_capacity);

/* The receiver of cepstrum_get_column, on a buffer instead of a file descriptor: scan for 0xa5
 * followed by 0x5a exactly like expect_magic (a 0xa5 not followed by 0x5a drops both bytes), then
 * take n_cep big-endian int16; repeat.  Stops at max_frames or when the buffer cannot hold another
 * whole column.  *n_frames_out = columns decoded, *consumed_out = bytes of `bytes` used up.   */
int  mfcc_hip_serial_unpack(const uint8_t *bytes, size_t n_bytes, int n_cep, int16_t *cep, size_t max_frames,
                            size_t *n_frames_out, size_t *consumed_out);

/* cepstrum_eval_power (software/cepstrum.c:161-183): `window` is the circular buffer of
 * n_frames x n_cep int16, `head` the element index of its oldest entry (0 for a linear window).
 * Sums the squares of the elements head + i, i = size/3, size/3 + n_cep, ... < 2 size/3 (size =
 * n_frames * n_cep; the first coefficient of the middle third of the frames when size/3 is a
 * multiple of n_cep, as in the reference's 16 x 93 window).  *power_out gets the sum (64-bit; the
 * reference accumulates in a 32-bit int); returns 1 if it reaches the reference's threshold 1e8,
 * 0 if not, negative on bad arguments.                                                       */
int  mfcc_hip_eval_power(const int16_t *window, int n_cep, int n_frames, size_t head, long long *power_out);

#ifdef __cplusplus
}
#endif
#endif /* MFCC_HIP_H */
