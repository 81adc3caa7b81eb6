"""The host emulator's own hazard detectors (tests/hostsim/hip_emu.h) against kernels with DELIBERATE hazards
(tests/hostsim/emu_selftest.hip): the default schedule and the immediate LDS-DMA let all three through -- as a lucky GPU run
would -- the reversed wave order, the depth-first schedule and the adversarial DMA timing each catch theirs, and the versions
with the missing barrier / wait in place pass under every mode.  The kernel cases and whole-model forwards of the suite run under
these modes (test_hostsim_kernels.py, test_engine_sim.py, test_gimmvfi_f.py)."""
import ctypes as C

import pytest

from sim_runtime import hostsim_lib


def _run(dll, which, fixed, src):
    out = (C.c_int * 64)()
    dll.gvfi_emu_selftest(which, fixed, src, out)
    if which == 1:      # the LDS-DMA case: lane l reads the first dword of its 16 bytes
        return all(out[l] == int.from_bytes(bytes(src[l * 16:l * 16 + 4]), "little", signed=True) for l in range(64))
    return list(out) == [i + 1 for i in range(64)]


@pytest.mark.parametrize("sched", [0, 1, 2, 3])
@pytest.mark.parametrize("dma", [0, 1])
def test_deliberate_hazards_are_caught_by_the_mode_made_for_them(sched, dma):
    dll = hostsim_lib().dll
    src = (C.c_ubyte * 1024)(*[(i * 7 + 3) & 255 for i in range(1024)])
    dll.gvfi_emu_set_sched(sched)
    dll.gvfi_emu_set_dma_mode(dma)
    try:
        got = {name: (_run(dll, which, 0, src), _run(dll, which, 1, src)) for name, which in (("race", 0), ("dma", 1), ("skew", 2))}
    finally:
        dll.gvfi_emu_set_sched(0)
        dll.gvfi_emu_set_dma_mode(1)      # (the emulator's default)
    # (hazard version passes?, fixed version passes?)
    assert got["race"] == (not (sched & 1), True), got       # no barrier between producer and consumer wave: reversed wave order
    assert got["skew"] == (sched != 2, True), got            # waves that must not drift apart: depth first (in thread order: in the
    #                                                          reversed order the PRODUCER is the wave that runs ahead -- every mode is ONE
    #                                                          interleaving, which is why the suite runs its kernel cases under several)
    assert got["dma"] == (dma == 0, True), got               # LDS read without the vmcnt wait: adversarial DMA timing


def test_random_schedules_find_the_missing_barrier_for_some_seed_and_never_fault_the_fixed_kernel():
    dll = hostsim_lib().dll
    src = (C.c_ubyte * 1024)(*[(i * 7 + 3) & 255 for i in range(1024)])
    caught = 0
    try:
        dll.gvfi_emu_set_sched(4)
        for seed in range(1, 9):
            dll.gvfi_emu_set_seed(seed)
            caught += not _run(dll, 0, 0, src)
            assert _run(dll, 0, 1, src) and _run(dll, 2, 1, src), seed
    finally:
        dll.gvfi_emu_set_sched(0)
        dll.gvfi_emu_set_seed(1)
    assert caught >= 2, caught
