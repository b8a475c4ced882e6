/*
 * vs_commguard.c -- see vs_commguard.h.  Plain C + pthreads, no device.
 *
 * Why not one mutex around every call (round 5 did that): a thread blocked INSIDE ncclSend / the root's receive group --
 * waiting for a peer that has failed and will never post its side -- would hold the very lock the failing thread needs
 * for ncclCommAbort, and the gather would hang instead of returning the error.  ncclCommAbort from another thread is how
 * a blocked call is ended (it is what watchdogs of collective libraries do); what must NOT happen is a NEW call entering
 * a communicator that is being torn down, or the pointer being forgotten while somebody still uses it -- that is all
 * this guard serialises.
 */
#include "vs_commguard.h"

#include <stddef.h>

void vs_commguard_init(VsCommGuard *g)
{
  pthread_mutex_init(&g->m, NULL);
  pthread_cond_init(&g->cv, NULL);
  g->comm = NULL;
  g->in_call = 0;
  g->dead = g->aborting = false;
}

void vs_commguard_destroy(VsCommGuard *g)
{
  pthread_cond_destroy(&g->cv);
  pthread_mutex_destroy(&g->m);
}

void vs_commguard_set(VsCommGuard *g, void *comm)
{
  pthread_mutex_lock(&g->m);
  g->comm = comm;
  g->dead = g->aborting = false;
  pthread_mutex_unlock(&g->m);
}

void *vs_commguard_enter(VsCommGuard *g)
{
  pthread_mutex_lock(&g->m);
  void *c = g->dead ? NULL : g->comm;
  if (c) g->in_call++;
  pthread_mutex_unlock(&g->m);
  return c;
}

void vs_commguard_leave(VsCommGuard *g)
{
  pthread_mutex_lock(&g->m);
  if (--g->in_call == 0) pthread_cond_broadcast(&g->cv);
  pthread_mutex_unlock(&g->m);
}

void vs_commguard_close(VsCommGuard *g)
{
  pthread_mutex_lock(&g->m);
  g->dead = true;
  pthread_mutex_unlock(&g->m);
}

int vs_commguard_abort(VsCommGuard *g, int (*abort_fn)(void *comm))
{
  pthread_mutex_lock(&g->m);
  void *c = g->aborting ? NULL : g->comm;
  g->dead = g->aborting = true;
  pthread_mutex_unlock(&g->m);
  if (!c) return 0;
  (void)abort_fn(c); /* no lock held: the threads blocked inside the communicator come back because of THIS */
  pthread_mutex_lock(&g->m);
  while (g->in_call > 0) pthread_cond_wait(&g->cv, &g->m);
  g->comm = NULL;
  pthread_mutex_unlock(&g->m);
  return 1;
}

void *vs_commguard_take(VsCommGuard *g)
{
  pthread_mutex_lock(&g->m);
  void *c = (!g->dead && g->in_call == 0) ? g->comm : NULL;
  if (c) g->comm = NULL;
  if (g->dead && g->in_call == 0) g->comm = NULL;
  pthread_mutex_unlock(&g->m);
  return c;
}
