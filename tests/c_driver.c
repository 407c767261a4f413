/*
 * c_driver.c -- a plain C caller of libmfcc_hip.so: the reference host driver's flow
 * (software/main.c: mfcc_open :36, mfcc_convert :100, mfcc_close :53, main :249) with the FT601 USB
 * ping-pong replaced by the C ABI of include/mfcc_hip.h, exactly as INTEGRATION.md section 1 shows it.
 * Built by __graft_entry__.build() (gcc, C99, no torch, no Python) and run as a child process by
 * tests/test_gpu_c_driver.py, which compares the `.mfcc` bytes it writes with the oracle.
 *
 *   c_driver convert  <in.wav> <out.mfcc>      fixed-point contract (what the FPGA would have written)
 *   c_driver convertf <in.wav> <out.mfcc>      float contract truncated to int16 (software/lift.py:39)
 *   c_driver stream   <in.wav> <out.mfcc> <chunk>   the same through the streaming session, `chunk` samples
 *                                                    per push (the driver's own transfer pattern: 512, then 170)
 *   c_driver lift     <in.mfcc> <out.lift>     software/lift.py:28-40 (host only)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfcc_hip.h"

#define NFFT        512          /* software/main.c:11-14 */
#define STEPSIZE    170
#define NCEPSTRUMS  32
#define SAMPLERATE  16000

struct mfcc_s {                  /* was: struct ft601_context ft601;  main.c:16-19 */
    mfcc_hip_handle *h;
    mfcc_hip_params p;
};

static int mfcc_open(struct mfcc_s *sess)                       /* main.c:36 */
{
    int ret;

    memset(sess, 0, sizeof *sess);
    mfcc_hip_default_params(&sess->p);
    sess->p.nfft = NFFT;
    sess->p.hop = STEPSIZE;
    sess->p.n_mel = 32;                                         /* mfcc/targets/wav2mfcc.py:19 */
    sess->p.n_cep = NCEPSTRUMS;
    sess->p.sample_rate = SAMPLERATE;
    sess->p.pad_mode = MFCC_HIP_PAD_STREAM;                     /* zero-padded tail frame, main.c:134-144 */
    ret = mfcc_hip_create(&sess->p, &sess->h);
    if (ret) {
        printf("Error mfcc_hip_create %d (%s), hip error %d\n", ret, mfcc_hip_strerror(ret),
               mfcc_hip_last_hip_error(NULL));
        return ret;
    }
    return 0;
}

static void mfcc_close(struct mfcc_s *sess)                     /* main.c:53 */
{
    mfcc_hip_destroy(sess->h);
    sess->h = NULL;
}

static int mfcc_convert(struct mfcc_s *sess, const char *path_in, const char *path_out, int fixed)   /* main.c:100 */
{
    size_t nframes = 0;
    int ret = mfcc_hip_convert_wav(sess->h, path_in, path_out, fixed, &nframes);

    if (ret) {
        printf("Error mfcc_hip_convert_wav %d (%s)\n", ret, mfcc_hip_strerror(ret));
        return -1;
    }
    printf("%s %s frames %lu kernel %s\n", path_in, path_out, (unsigned long)nframes,
           mfcc_hip_kernel_name(sess->h, fixed));
    return 0;
}

/* 16-bit mono PCM: the canonical 44-byte header is enough for the driver test's own files */
static int16_t *read_wav(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    unsigned char hdr[44];
    long size;
    int16_t *pcm;

    if (!f) return NULL;
    if (fread(hdr, 1, 44, f) != 44 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 36, "data", 4)) {
        fclose(f);
        return NULL;
    }
    fseek(f, 0, SEEK_END);
    size = ftell(f) - 44;
    fseek(f, 44, SEEK_SET);
    pcm = malloc((size_t)size + 2);
    *n = fread(pcm, 2, (size_t)size / 2, f);
    fclose(f);
    return pcm;
}

/* the per-frame loop of mfcc_convert (main.c:128-166) on the streaming session: first NFFT samples, then
 * STEPSIZE per round, one column of NCEPSTRUMS back per round; zero padding at EOF = flush */
static int mfcc_convert_stream(struct mfcc_s *sess, const char *path_in, const char *path_out, size_t chunk)
{
    mfcc_hip_stream *st = NULL;
    size_t n = 0, pos = 0, got = 0, total = 0, cap;
    int16_t *pcm = read_wav(path_in, &n);
    int16_t *cep;
    FILE *out;
    int ret;

    if (!pcm) {
        printf("Failed to read %s\n", path_in);
        return -1;
    }
    ret = mfcc_hip_stream_create(sess->h, /*fixed=*/1, &st);
    if (ret) {
        printf("Error mfcc_hip_stream_create %d (%s)\n", ret, mfcc_hip_strerror(ret));
        free(pcm);
        return -1;
    }
    out = fopen(path_out, "wb");
    if (!out) {
        mfcc_hip_stream_destroy(st);
        free(pcm);
        return -1;
    }
    cap = (chunk / STEPSIZE + 4) * NCEPSTRUMS;
    cep = malloc(cap * sizeof *cep);
    mfcc_hip_stream_reset(st);                                  /* mfcc_softreset(sess), main.c:113 */
    while (pos < n) {
        size_t amount = chunk ? chunk : (pos == 0 ? NFFT : STEPSIZE);       /* main.c:134 */
        if (amount > n - pos) amount = n - pos;
        ret = mfcc_hip_stream_push(st, pcm + pos, amount, cep, cap, &got);
        if (ret) {
            printf("Error mfcc_hip_stream_push %d (%s)\n", ret, mfcc_hip_strerror(ret));
            goto done;
        }
        fwrite(cep, sizeof *cep, got * NCEPSTRUMS, out);        /* main.c:162-165 */
        total += got;
        pos += amount;
    }
    ret = mfcc_hip_stream_flush(st, cep, cap, &got);            /* the zero-padded round that hits EOF */
    if (ret) {
        printf("Error mfcc_hip_stream_flush %d (%s)\n", ret, mfcc_hip_strerror(ret));
        goto done;
    }
    fwrite(cep, sizeof *cep, got * NCEPSTRUMS, out);
    total += got;
    printf("%s %s frames %lu (streamed, %lu samples per push)\n", path_in, path_out, (unsigned long)total,
           (unsigned long)chunk);
done:
    fclose(out);
    free(cep);
    free(pcm);
    mfcc_hip_stream_destroy(st);
    return ret ? -1 : 0;
}

int main(int argc, char *argv[])
{
    struct mfcc_s sess;
    int ret;

    if (argc < 4) {
        printf("Usage: %s convert|convertf|stream|lift <in> <out> [chunk]\n", argv[0]);
        return 1;
    }
    if (!strcmp(argv[1], "lift")) {
        size_t nframes = 0;
        ret = mfcc_hip_lift_file(argv[2], argv[3], NCEPSTRUMS, 22.0, &nframes);
        printf("%s %s cepstrum sets: %lu\n", argv[2], argv[3], (unsigned long)nframes);
        return ret ? 2 : 0;
    }
    ret = mfcc_open(&sess);
    if (ret) return 3;
    if (!strcmp(argv[1], "convert")) ret = mfcc_convert(&sess, argv[2], argv[3], 1);
    else if (!strcmp(argv[1], "convertf")) ret = mfcc_convert(&sess, argv[2], argv[3], 0);
    else if (!strcmp(argv[1], "stream")) ret = mfcc_convert_stream(&sess, argv[2], argv[3], argc > 4 ? (size_t)atol(argv[4]) : 0);
    else ret = -1;
    mfcc_close(&sess);
    return ret ? 4 : 0;
}
