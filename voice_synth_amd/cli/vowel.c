/*
 * vowel -- drop-in for the reference's vocal-tract filter program, computing on MI355X.
 *
 * Same command line, stdout text, exit codes and output file as /root/reference/vowel_new.c
 * (option loop :116-192, file handling :194-219, msg() :404-410, usage() :345-355).  The
 * per-sample loop (:252-296) runs as one lane of the batched gfx950 kernel through
 * vs_filter(); the frame chunking of :237 and overlap() :392-400 only structure the
 * reference's file I/O (the filter state persists across frames), so the whole
 * payload is filtered in one call.
 * "-n" (white noise added to the filtered signal frame by frame, vowel_new.c:302-324) runs on
 * the device too: lane.out_snr / lane.out_seed.
 */
#include <math.h>

#include "cli_common.h"

static void usage(void)
{
  /* text of vowel_new.c:347-353 */
  printf("lxfilter -i file1.wav  -o file2.wav [- arg {descr (defaults)}]\n");
  printf("-i   {input file  (.wav, pcm, mono, 16 bits/sample, 22050 Hz)}\n");
  printf("-o   {output file (.wav, pcm, mono, 16 bits/sample, 22050 Hz)}\n");
  printf("-v x {vowel: n = x, i, or u}\n");
  printf("-p x {pre_emphasis (0.0 <= x <= 1.0, 1.0}\n");
  printf("-g x {gain (x > 0.0, 10.0)}\n");
  printf("-n x {SNR ratio (white noise added to oral pressure), x > 0}\n");
  exit(0);
}

int main(int argc, char **argv)
{
  vs_vowel_cmd cmd;
  int rc = vs_vowel_parse(argc, argv, &cmd);
  if (rc == VS_USAGE) usage();
  if (rc != VS_OK) {
    fprintf(stderr, "vowel: %s\n", vs_strerror(rc));
    return 1;
  }
  if (cmd.output_arg == -1) usage(); /* the reference would fopen(argv[-1]) */

  vs_lane lane;
  vs_lane_defaults(&lane);
  lane.gain = cmd.gain;
  lane.pre_emphasis = cmd.pre_emphasis;
  lane.vowel = cmd.vowel;
  if (cmd.noise_arg != -1) {
    lane.out_snr = cmd.snr;         /* already pow(10, x/10), vowel_new.c:143 */
    lane.out_seed = vs_cli_seed();  /* replaces srandom(time(NULL)), vowel_new.c:234 */
  }
  rc = vs_lane_validate(&lane);
  if (rc != VS_OK) {
    fprintf(stderr, "vowel: -v %c: %s\n", (char)cmd.vowel, vs_strerror(rc));
    return 1;
  }
  /* coefficients() prints the table label while the options are parsed, vowel_new.c:550-622 */
  printf("vowel %s", vs_vowel_name(cmd.vowel));

  /* input file, vowel_new.c:195-210 */
  FILE *fdr = fopen(argv[cmd.input_arg], "rb");
  if (fdr == NULL) {
    printf(".wav file not found\n");
    exit(0);
  }
  unsigned char header[72];
  size_t got = fread(header, 1, sizeof(header), fdr);
  int32_t fs = 0;
  int tag = 0, bits = 0;
  uint64_t data_bytes = 0;
  int hbytes = vs_wav_header_read(header, got, &fs, &tag, &bits, &data_bytes);
  if (hbytes < 0) {
    printf(".wav file is not PCM"); /* what the reference reports for an unrecognised header */
    exit(0);
  }
  if (tag != 1) {
    printf(".wav file is not PCM");
    exit(0);
  }
  if (bits != 16) {
    printf(".wav file is not 16 bits per sample!");
  }

  /* payload: everything after the header, in whole samples (the reference reads until EOF) */
  if (fseek(fdr, 0, SEEK_END) != 0) return 1;
  long fsize = ftell(fdr);
  size_t n = (fsize > hbytes) ? (size_t)(fsize - hbytes) / sizeof(int16_t) : 0;
  fseek(fdr, hbytes, SEEK_SET);
  int16_t *x = (int16_t *)malloc((n ? n : 1) * sizeof(int16_t));
  int16_t *y = (int16_t *)malloc((n ? n : 1) * sizeof(int16_t));
  if (!x) {
    printf("out of memory in call to malloc(x).\n");
    exit(1);
  }
  if (!y) {
    printf("out of memory in call to malloc(y).\n");
    exit(1);
  }
  if (n && fread(x, sizeof(int16_t), n, fdr) != n) n = 0;

  /* output file with the input's header copied verbatim, vowel_new.c:213-219 */
  FILE *fpwb = fopen(argv[cmd.output_arg], "wb");
  if (fpwb == NULL) {
    printf("Error while creating file (%s)\n", argv[cmd.output_arg]);
    exit(1);
  }
  fwrite(header, (size_t)hbytes, 1, fpwb);

  /* msg(), vowel_new.c:404-410, and the parameter line :231 */
  printf(" \nMaurilio N. Vieira, 28 mar 97. \n");
  printf(" Cascade Formant Synthesiser\n");
  printf(" Formant frequencies/bandwithds from Rabiner & Schafer (1978),\n");
  printf(" Digital Processing of Speech Signals, Prentice Hall, pp. 74-77\n");
  printf("pre_emphasis=%5.2f, gain=%5.2f, snr=%5.2f\n", cmd.pre_emphasis, cmd.gain, cmd.snr);
  printf("Wait...");

  if (n) {
    lane.fs = fs; /* the frame length of -n follows the input file's rate, vowel_new.c:361 */
    vs_ctx *ctx = NULL;
    if (vs_cli_open_ctx(&ctx) != VS_OK) return 1;
    rc = vs_filter(ctx, &lane, 1, n, x, y);
    if (rc != VS_OK) {
      fprintf(stderr, "\nvowel: filtering failed: %s (hip %d)\n", vs_strerror(rc),
              vs_ctx_last_hip_error(ctx));
      return 1;
    }
    vs_ctx_destroy(ctx);
    fwrite(y, sizeof(int16_t), n, fpwb);
  }

  free(x);
  free(y);
  fclose(fdr);
  fclose(fpwb);
  printf("done\n");
  exit(0);
}
