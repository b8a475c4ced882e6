"""Host-side C of the product (voice_synth_amd/csrc/vs_host.c, vs_planhost.c: everything of the host side that needs
no device) and the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU builds only: GPU sanitizers are not
available on this pool."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off"]


def test_host_c_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "test_host_asan")
    subprocess.run(["gcc"] + SAN + [os.path.join(ROOT, "tests", "c", "test_host_asan.c"),
                                    os.path.join(ROOT, "voice_synth_amd", "csrc", "vs_host.c"),
                                    "-I" + os.path.join(ROOT, "include"), "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.strip() == b"ok"


def test_plan_host_c_under_asan_ubsan(tmp_path):
    """the device-free half of plan creation: expansion on 8 threads, the stable order, the ring policy, the gather
    rounds (tests/c/test_planhost_asan.c)"""
    exe = str(tmp_path / "test_planhost_asan")
    csrc = os.path.join(ROOT, "voice_synth_amd", "csrc")
    subprocess.run(["gcc", "-std=gnu11"] + SAN + [os.path.join(ROOT, "tests", "c", "test_planhost_asan.c"),
                                                  os.path.join(csrc, "vs_planhost.c"), os.path.join(csrc, "vs_host.c"),
                                                  "-I" + os.path.join(ROOT, "include"), "-o", exe, "-lm", "-lpthread"], check=True)
    r = subprocess.run([exe], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.strip() == b"ok"


def test_communicator_guard_ends_a_blocked_exchange(tmp_path):
    """csrc/vs_commguard.c, the guard around the node's RCCL communicators, against a mock communicator whose calls block
    until their peer answers (tests/c/test_commguard.c): a shard that fails before it posts its side must be able to
    abort the exchange while its peers -- and the root -- are blocked inside their calls.  Under ASan/UBSan, and under
    ThreadSanitizer where the toolchain has it."""
    src = [os.path.join(ROOT, "tests", "c", "test_commguard.c"), os.path.join(ROOT, "voice_synth_amd", "csrc", "vs_commguard.c")]
    for name, flags in (("asan", SAN), ("tsan", ["-g", "-O1", "-fsanitize=thread"])):
        exe = str(tmp_path / ("test_commguard_" + name))
        built = subprocess.run(["gcc", "-std=gnu11"] + flags + src + ["-o", exe, "-lpthread"], capture_output=True)
        if built.returncode != 0 and name == "tsan":
            continue                                  # no libtsan here: the ASan run stands
        assert built.returncode == 0, built.stderr
        r = subprocess.run([exe], capture_output=True, timeout=120)
        assert r.returncode == 0 and r.stdout.strip() == b"ok", (name, r.stdout, r.stderr)


ORACLE_DRIVER = r"""
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle/vs_oracle.h"
int main(void) {
  vs_lane l; memset(&l, 0, sizeof l);
  l.cq = 0.55f; l.K = 0.65f; l.Fg = 125; l.F0 = 120; l.fs = 16000; l.amp = 12000; l.gain = 10; l.pre_emphasis = 1;
  l.vowel = '1'; l.A[0] = 1.0;
  l.jitter = 0.03f; l.shimmer = 0.1f; l.noise = 10.0f; l.DC = 1200.0f; l.Kvar = 0.5f;
  l.flags = VS_FLAG_JITTER | VS_FLAG_SHIMMER | VS_FLAG_NOISE; l.seed = 5; l.out_snr = 100.0f; l.out_seed = 6;
  size_t n = 16000; int16_t *f = malloc(n * 2), *p = malloc(n * 2); vs_cycle_rec rec[400]; int32_t nc; uint64_t nd;
  if (vs_oracle_source(&l, n, f, rec, 400, &nc, &nd)) return 2;
  if (vs_oracle_filter(&l, n, f, p)) return 3;
  l.amp = 32766; l.DC = 0.25f; l.shimmer = 0.5f;             /* wraps the short conversion */
  if (vs_oracle_source(&l, n, f, rec, 400, &nc, &nd)) return 4;
  l.fs = 44100; l.F0 = 50; l.Fg = 52; l.jitter = 0.1f;       /* the case that overflows the reference's w[500] */
  n = 30000; free(f); free(p); f = malloc(n * 2); p = malloc(n * 2);
  if (vs_oracle_source(&l, n, f, NULL, 0, &nc, &nd)) return 5;
  if (vs_oracle_filter(&l, n, f, p)) return 6;
  free(f); free(p); printf("ok\n"); return 0;
}
"""


def test_oracle_under_asan_ubsan(tmp_path):
    src = tmp_path / "drv.c"
    src.write_text(ORACLE_DRIVER)
    exe = str(tmp_path / "drv")
    subprocess.run(["gcc"] + SAN + ["-fno-sanitize=float-cast-overflow", str(src),
                                    os.path.join(ROOT, "oracle", "vs_oracle.c"), "-I" + ROOT,
                                    "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
