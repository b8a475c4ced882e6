/*
 * flowgen_shimmer -- drop-in for the reference's glottal-source program, computing on MI355X.
 *
 * Same command line, stdout text, exit codes and output file as /root/reference/
 * flowgen_shimmer.c (option loop :128-219, initialization() :463-574, msg() :576-588,
 * usage() :436-461, per-cycle prints :307 and :409).  The sample loop (:246-423) runs as one
 * lane of the batched gfx950 kernel through vs_source(); this program holds no arithmetic of
 * its own beyond formatting the per-cycle diagnostics the kernel returns.
 */
#include <math.h>

#include "cli_common.h"

static void usage(void)
{
  /* text of flowgen_shimmer.c:438-458; scripts may grep it */
  printf("(c) Maurilio N. Vieira, 28 mar 1997\n");
  printf(" Simulated airflow based on Fant (1979),\n");
  printf(" Glottal Source and Excitation Analysis,\n");
  printf(" STL-QPSR 1/1979, pp. 85-107\n\n");
  printf("usage:\n");
  printf("%s -o file [-args {description (defaults <range>)}]\n\n", "voicegen");
  printf("-o x {Output file (.wav, pcm, 16 bits/sample)}\n");
  printf("-r x {sampling Rate (22050 Hz <44100, 22050, or 11025>)}\n");
  printf("-d x {Duration (> 0.5 seconds)}\n");
  printf("-j x {jitter (0%% <0-10%%>)}\n");
  printf("-c x {closed quotient (.55 <0-1>)}\n");
  printf("-f x {Fundamental frequency, F0, 120 Hz }\n");
  printf("-g x {Glottal formant, Fg > F0, in Fant's (1979) model, 125 Hz }\n");
  printf("-k x {Speed of closure, K, in Fant's (1979) model 0.65 <0.55-1.00>}\n");
  printf("-z x {Variation of speed of closure (0.0 <0-1.0>))}\n");
  printf("-s x {shimmer (0.0 <0-10%%>))}\n");
  printf("-n x {cycle-to-cycle SNR (0 dB  <0-50>) \n");
  printf("      aditive noise, uniforme distribution, closed phase}\n");
  printf("-a x {maximum amplitude (12000 <0-32767>)}\n");
  printf("-l x {DC flow, proportion of max amplitude (0.0 <0-0.30>))}\n");
  exit(0);
}

int main(int argc, char **argv)
{
  vs_flowgen_cmd cmd;
  int rc = vs_flowgen_parse(argc, argv, &cmd);
  if (rc == VS_USAGE) usage();
  if (rc != VS_OK) {
    fprintf(stderr, "flowgen_shimmer: %s\n", vs_strerror(rc));
    return 1;
  }
  vs_lane *par = &cmd.lane;
  const char *path = argv[cmd.wav_arg];
  const int noise_on = (par->flags & VS_FLAG_NOISE) != 0;
  const int shimmer_on = (par->flags & VS_FLAG_SHIMMER) && par->shimmer != 0.0;

  rc = vs_lane_validate(par);
  if (rc != VS_OK) {
    fprintf(stderr, "flowgen_shimmer: %s\n", vs_strerror(rc));
    return 1;
  }

  /* header, fg:550-565 */
  unsigned char header[72];
  int hbytes = vs_wav_header_write(header, vs_cli_header_bytes(), par->fs, cmd.dur);

  FILE *outfile = fopen(path, "wb");
  if (outfile == NULL) {
    printf("Error while creating %s\n", path); /* fg:226; the reference then crashes */
    return 1;
  }
  if (fwrite(header, (size_t)hbytes, 1, outfile) != 1) {
    printf("Error while writing header to %s\n", path);
    exit(0); /* fg:232 */
  }

  /* msg(), fg:576-588 */
  printf("(c) Maurilio N. Vieira, 1996\nSynthetic vowel generator\n");
  printf("ported to gcc - Joao SANSAO, Feb. 2007");
  printf("Output file = %s\n", path);
  if (noise_on) printf("SNR: %5.2f dB, ", 10.0 * log10(par->noise));
  printf("Fs=%ld Hz, Dur=%5.2f s, Fg=%d Hz, Amp = %d, DCflow=%5.2f\n", (long)par->fs, cmd.dur,
         (int)par->Fg, par->amp, par->DC);
  printf("Wait...");

  uint64_t n_samples = 0;
  rc = vs_num_samples(par->fs, cmd.dur, &n_samples); /* fg:242 */
  if (rc != VS_OK) { /* "-d inf": the reference's conversion of the product is undefined and its loop endless */
    fprintf(stderr, "\nflowgen_shimmer: cannot synthesise %g s at %ld Hz: %s\n", (double)cmd.dur, (long)par->fs, vs_strerror(rc));
    return 1;
  }
  par->seed = vs_cli_seed();                    /* replaces srandom(time(NULL)), fg:241 */

  vs_ctx *ctx = NULL;
  if (vs_cli_open_ctx(&ctx) != VS_OK) return 1;

  const int P = (int)((float)par->fs / par->F0);
  size_t max_cyc = (size_t)(n_samples / (uint64_t)(P > 2 ? (P * 4) / 5 : 1)) + 8;
  int16_t *x = (int16_t *)malloc((size_t)n_samples * sizeof(int16_t));
  vs_cycle_rec *recs = (vs_cycle_rec *)calloc(max_cyc, sizeof(vs_cycle_rec));
  int32_t ncyc = 0;
  if (!x || !recs) {
    printf("out of memory in call to malloc(x).\n"); /* fg:570 */
    exit(1);
  }
  rc = vs_source(ctx, par, 1, (size_t)n_samples, x, recs, max_cyc, &ncyc);
  if (rc != VS_OK) {
    fprintf(stderr, "\nflowgen_shimmer: synthesis failed: %s (hip %d)\n", vs_strerror(rc),
            vs_ctx_last_hip_error(ctx));
    return 1;
  }

  /* per-cycle diagnostics in the reference's order: S (fg:307) then SNRdb (fg:409) */
  for (int32_t c = 0; c < ncyc && (size_t)c < max_cyc; c++) {
    if (shimmer_on) printf("%5.2f \n", recs[c].S);
    if (noise_on) printf("SNRdb = %5.2f\n", 10.0 * log10(recs[c].x_pow / recs[c].w_pow));
  }

  if (fwrite(x, sizeof(int16_t), (size_t)n_samples, outfile) != (size_t)n_samples) {
    printf("Error while writing samples to %s\n", path);
    exit(0); /* fg:420 */
  }
  free(x);
  free(recs);
  fclose(outfile);
  vs_ctx_destroy(ctx);
  printf("done\n");
  exit(0);
}
