/*
 * oracle/ref_pipelines.c -- times the REFERENCE AS SHIPPED over many utterances, all host cores.
 *
 * TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).  The reference is two programs that talk
 * through a .wav file (/root/reference/Makefile:3-13 builds them, no -O flag); what a user of it does
 * for N utterances is N times "flowgen_shimmer -o f.wav ...; vowel -i f.wav -o v.wav ...", e.g. under
 * xargs -P $(nproc).  This helper does exactly that with posix_spawn, without a shell or an
 * interpreter in the loop: `workers` worker processes, each in a scratch directory of its own, each
 * running its share of the manifest's pipelines one after the other.  (The round-2 harness ran the
 * pipelines from a Python thread pool inside the torch process and measured the launcher.)
 *
 *   ref_pipelines <flowgen-binary> <vowel-binary> <manifest> <workers> <scratch-dir>
 *
 * manifest: one utterance per line, "<seed>|<flowgen arguments>|<vowel arguments>" (arguments separated
 * by blanks, without -o / -i: the helper adds "-o f.wav" and "-i f.wav -o v.wav").  VS_SEED=<seed> is
 * set for both programs (oracle/rng_shim.c).  stdout of the programs goes to /dev/null.
 * Prints one JSON line: pipelines done, failures, wall seconds, the time spent inside each of the two
 * programs (spawn to exit, summed over workers), bytes of the last v.wav of every worker (a check that
 * the programs really ran).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <spawn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

extern char **environ;

#define MAX_ARGS 64

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* splits s at blanks into argv[n..], returns the new n */
static int split_args(char *s, char **argv, int n)
{
  char *save = NULL;
  for (char *t = strtok_r(s, " \t\r\n", &save); t && n < MAX_ARGS - 1; t = strtok_r(NULL, " \t\r\n", &save)) argv[n++] = t;
  return n;
}

static int run(const char *prog, char **argv, char **envp, posix_spawn_file_actions_t *fa)
{
  pid_t pid;
  if (posix_spawn(&pid, prog, fa, NULL, argv, envp) != 0) return -1;
  int status = 0;
  while (waitpid(pid, &status, 0) < 0) {
    if (errno != EINTR) return -1;
  }
  return (WIFEXITED(status) && WEXITSTATUS(status) == 0) ? 0 : -1;
}

int main(int argc, char **argv)
{
  if (argc != 6) {
    fprintf(stderr, "usage: ref_pipelines <flowgen> <vowel> <manifest> <workers> <scratch-dir>\n");
    return 2;
  }
  const char *flowgen = argv[1], *vowel = argv[2], *manifest = argv[3], *scratch = argv[5];
  int workers = atoi(argv[4]);
  if (workers < 1) workers = 1;

  /* the manifest, once, in the parent */
  FILE *mf = fopen(manifest, "r");
  if (!mf) {
    perror(manifest);
    return 2;
  }
  size_t cap = 1024, n_lines = 0;
  char **lines = (char **)malloc(cap * sizeof(char *));
  char *buf = NULL;
  size_t blen = 0;
  while (getline(&buf, &blen, mf) > 0) {
    if (buf[0] == '\n' || buf[0] == '#') continue;
    if (n_lines == cap) lines = (char **)realloc(lines, (cap *= 2) * sizeof(char *));
    lines[n_lines++] = strdup(buf);
  }
  fclose(mf);
  free(buf);
  if ((size_t)workers > n_lines) workers = (int)(n_lines ? n_lines : 1);

  /* the programs' environment: ours + VS_SEED (slot filled per pipeline) */
  int n_env = 0;
  while (environ[n_env]) n_env++;
  char **envp = (char **)calloc((size_t)n_env + 2, sizeof(char *));
  int e = 0;
  for (int i = 0; i < n_env; i++)
    if (strncmp(environ[i], "VS_SEED=", 8) != 0 && strncmp(environ[i], "VS_DRAWLOG=", 11) != 0) envp[e++] = environ[i];
  const int seed_slot = e;
  envp[e + 1] = NULL;

  int pipes[2];
  if (pipe(pipes) != 0) return 2;
  const double t0 = now_s();
  for (int w = 0; w < workers; w++) {
    const pid_t pid = fork();
    if (pid < 0) return 2;
    if (pid == 0) {
      close(pipes[0]);
      char dir[512];
      snprintf(dir, sizeof(dir), "%s/w%d", scratch, w);
      mkdir(dir, 0700);
      if (chdir(dir) != 0) _exit(3); /* short file names: the reference's WaveFile[30], SURVEY F16 */
      posix_spawn_file_actions_t fa;
      posix_spawn_file_actions_init(&fa);
      posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
      long done = 0, failed = 0;
      double t_flow = 0.0, t_vowel = 0.0;
      for (size_t i = (size_t)w; i < n_lines; i += (size_t)workers) {
        char *line = lines[i];
        char *bar1 = strchr(line, '|');
        char *bar2 = bar1 ? strchr(bar1 + 1, '|') : NULL;
        if (!bar1 || !bar2) {
          failed++;
          continue;
        }
        *bar1 = *bar2 = '\0';
        char seedvar[64];
        snprintf(seedvar, sizeof(seedvar), "VS_SEED=%s", line);
        envp[seed_slot] = seedvar;
        char *av[MAX_ARGS];
        int n = 0;
        av[n++] = (char *)"flowgen_shimmer";
        av[n++] = (char *)"-o";
        av[n++] = (char *)"f.wav";
        n = split_args(bar1 + 1, av, n);
        av[n] = NULL;
        double ta = now_s();
        int rc = run(flowgen, av, envp, &fa);
        t_flow += now_s() - ta;
        if (rc == 0) {
          n = 0;
          av[n++] = (char *)"vowel";
          av[n++] = (char *)"-i";
          av[n++] = (char *)"f.wav";
          av[n++] = (char *)"-o";
          av[n++] = (char *)"v.wav";
          n = split_args(bar2 + 1, av, n);
          av[n] = NULL;
          ta = now_s();
          rc = run(vowel, av, envp, &fa);
          t_vowel += now_s() - ta;
        }
        if (rc == 0) done++;
        else failed++;
      }
      struct stat st;
      long bytes = (stat("v.wav", &st) == 0) ? (long)st.st_size : 0;
      unlink("f.wav");
      unlink("v.wav");
      if (chdir("/") == 0) rmdir(dir);
      double rep[5] = {(double)done, (double)failed, (double)bytes, t_flow, t_vowel};
      if (write(pipes[1], rep, sizeof(rep)) != (ssize_t)sizeof(rep)) _exit(4);
      _exit(0);
    }
  }
  close(pipes[1]);
  long done = 0, failed = 0, bytes = 0;
  double rep[5], t_flow = 0.0, t_vowel = 0.0;
  int reports = 0;
  while (read(pipes[0], rep, sizeof(rep)) == (ssize_t)sizeof(rep)) {
    done += (long)rep[0];
    failed += (long)rep[1];
    bytes += (long)rep[2];
    t_flow += rep[3];
    t_vowel += rep[4];
    reports++;
  }
  while (wait(NULL) > 0) {
  }
  const double t = now_s() - t0;
  /* process_seconds_*: spawn-to-exit time of the two programs summed over all pipelines and workers */
  printf("{\"pipelines\": %ld, \"failed\": %ld, \"workers\": %d, \"workers_reported\": %d, \"seconds\": %.4f, "
         "\"process_seconds_flowgen\": %.3f, \"process_seconds_vowel\": %.3f, \"last_wav_bytes\": %ld}\n",
         done, failed, workers, reports, t, t_flow, t_vowel, bytes);
  return (failed == 0 && reports == workers) ? 0 : 1;
}
