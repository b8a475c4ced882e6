/* The guard around the node's RCCL communicators (csrc/vs_commguard.c) against a MOCK communicator whose point-to-point
 * calls block on the host until their peer posts its side -- or until the communicator is aborted, which is how
 * ncclSend / ncclRecv behave while a connection is being set up.  Three shards: shard 1 sends and the root receives, both
 * blocked because shard 2 FAILS before it ever posts its send; shard 2 then ends the exchange the way
 * vs_node.c::abort_exchange does (close all, abort all).  With the round-5 scheme -- one mutex held across the call --
 * this test hangs (the failing thread waits for the lock of a communicator whose holder waits for the failed thread):
 * it must finish, every blocked call must come back with the abort's error, nobody may enter afterwards, and the
 * orderly path (nobody fails) must still work.  Runs under ASan/UBSan and under TSan (tests/test_host_sanitizers.py). */
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include "../../voice_synth_amd/csrc/vs_commguard.h"

typedef struct MockComm {
  pthread_mutex_t m;
  pthread_cond_t cv;
  int posted;    /* sides of the exchange that have been posted */
  int needed;    /* a call returns once this many are */
  bool aborted;
  bool freed;    /* "use after free" detector: set by the abort once every caller the mock knows of has returned */
  int inside;
} MockComm;

static int mock_call(MockComm *c)
{
  pthread_mutex_lock(&c->m);
  if (c->freed) abort(); /* somebody entered a communicator that is gone */
  c->inside++;
  c->posted++;
  pthread_cond_broadcast(&c->cv);
  while (c->posted < c->needed && !c->aborted) pthread_cond_wait(&c->cv, &c->m);
  const int rc = c->aborted ? 7 : 0;
  c->inside--;
  pthread_cond_broadcast(&c->cv);
  pthread_mutex_unlock(&c->m);
  return rc;
}
static int mock_abort(void *p)
{
  MockComm *c = (MockComm *)p;
  pthread_mutex_lock(&c->m);
  c->aborted = true;
  pthread_cond_broadcast(&c->cv);
  pthread_mutex_unlock(&c->m);
  return 0;
}

#define S 3
static VsCommGuard guard[S];
static MockComm comm[S];
static atomic_int came_back[S];
static int fail_shard = -1;

static void end_exchange(void)
{
  for (int p = 0; p < S; p++) vs_commguard_close(&guard[p]);
  for (int p = 0; p < S; p++) (void)vs_commguard_abort(&guard[p], mock_abort);
}

static void *shard(void *arg)
{
  const int s = (int)(long)arg;
  if (s == fail_shard) {
    usleep(100 * 1000); /* let the others get stuck first */
    end_exchange();
    atomic_store(&came_back[s], 2);
    return NULL;
  }
  MockComm *c = (MockComm *)vs_commguard_enter(&guard[s]);
  if (!c) {
    atomic_store(&came_back[s], 3);
    return NULL;
  }
  const int rc = mock_call(c); /* no lock of the guard held */
  vs_commguard_leave(&guard[s]);
  atomic_store(&came_back[s], rc == 0 ? 1 : 2);
  return NULL;
}

static int run(int failing)
{
  fail_shard = failing;
  for (int s = 0; s < S; s++) {
    pthread_mutex_init(&comm[s].m, NULL);
    pthread_cond_init(&comm[s].cv, NULL);
    comm[s].posted = 0;
    comm[s].needed = 1; /* every call of the orderly run is answered at once ... */
    comm[s].aborted = comm[s].freed = false;
    comm[s].inside = 0;
    vs_commguard_set(&guard[s], &comm[s]);
    atomic_store(&came_back[s], 0);
  }
  if (failing >= 0)
    for (int s = 0; s < S; s++) comm[s].needed = 2; /* ... the failing shard's side never comes */
  pthread_t t[S];
  for (int s = 0; s < S; s++) pthread_create(&t[s], NULL, shard, (void *)(long)s);
  for (int s = 0; s < S; s++) pthread_join(t[s], NULL);
  int bad = 0;
  for (int s = 0; s < S; s++) {
    const int want = failing < 0 ? 1 : 2;
    if (atomic_load(&came_back[s]) != want) bad++;
    if (failing >= 0) {
      if (vs_commguard_enter(&guard[s]) != NULL) bad++;   /* nobody enters a communicator that has been aborted */
      if (vs_commguard_take(&guard[s]) != NULL) bad++;    /* ... and there is nothing left to destroy */
      if (vs_commguard_abort(&guard[s], mock_abort) != 0) bad++; /* once per communicator */
    } else {
      if (vs_commguard_take(&guard[s]) != &comm[s]) bad++; /* the orderly path hands it back for ncclCommDestroy */
      if (vs_commguard_enter(&guard[s]) != NULL) bad++;
    }
    if (comm[s].inside != 0) bad++;
    pthread_cond_destroy(&comm[s].cv);
    pthread_mutex_destroy(&comm[s].m);
  }
  return bad;
}

int main(void)
{
  alarm(20); /* a hang is a failure */
  for (int s = 0; s < S; s++) vs_commguard_init(&guard[s]);
  int bad = run(-1);
  for (int rep = 0; rep < 20; rep++) bad += run(2) + run(1) + run(0);
  bad += run(-1);
  for (int s = 0; s < S; s++) vs_commguard_destroy(&guard[s]);
  if (bad) {
    printf("FAILED %d\n", bad);
    return 1;
  }
  printf("ok\n");
  return 0;
}
