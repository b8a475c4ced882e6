/*
 * vs_commguard.h -- who may be inside a communicator, and how an exchange that cannot complete is ended
 * (csrc/vs_commguard.c; used by the node's RCCL transport, csrc/vs_node.c).  No device, no RCCL in it:
 * tests/c/test_commguard.c runs it against a communicator whose calls block until it is aborted.
 */
#ifndef VS_COMMGUARD_H
#define VS_COMMGUARD_H

#include <pthread.h>
#include <stdbool.h>

typedef struct VsCommGuard {
  pthread_mutex_t m;
  pthread_cond_t cv;
  void *comm;   /* the communicator (opaque here); NULL: none, or gone */
  int in_call;  /* threads between vs_commguard_enter() and vs_commguard_leave() */
  bool dead;    /* nobody enters any more */
  bool aborting; /* some thread has taken the abort on */
} VsCommGuard;

void vs_commguard_init(VsCommGuard *g);
void vs_commguard_destroy(VsCommGuard *g);
/* a fresh communicator (or NULL) -- only while no thread can be inside the old one */
void vs_commguard_set(VsCommGuard *g, void *comm);
/* Before a host call into the communicator: returns it and counts the caller in, or NULL when there is none or it has
 * been aborted (the caller stops, without an error of its own).  The call itself is made WITHOUT any lock held: a
 * point-to-point call may block on the host until its peer answers (the first one sets the connection up), and a lock
 * held across it would keep the abort that is meant to end it from ever starting. */
void *vs_commguard_enter(VsCommGuard *g);
void vs_commguard_leave(VsCommGuard *g);
/* Nobody enters any more (callers that are inside stay inside).  An exchange over several communicators closes ALL of
 * them before it aborts the first: a thread that comes back from a call on one must not walk into the next. */
void vs_commguard_close(VsCommGuard *g);
/* Ends the communicator: closes it, calls abort_fn(comm) with no lock held -- that is what makes the calls other
 * threads are blocked in return --, waits until they have left and forgets the pointer.  Once per communicator, whoever
 * comes first; returns 1 if this call did it, 0 if there was nothing (left) to abort. */
int vs_commguard_abort(VsCommGuard *g, int (*abort_fn)(void *comm));
/* For an orderly destroy: the communicator if it is alive and nobody is inside (and forgets it), else NULL. */
void *vs_commguard_take(VsCommGuard *g);

#endif
