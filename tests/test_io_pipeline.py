"""Host I/O pipeline of the CLI: ordering, decode-once, windowing, error propagation.
Device-free here (CPU tensors); the CUDA streams / pinned buffers are exercised by the CLI test on the GPU."""
import threading
import time

import pytest
import torch

from gimmvfi_hip.io_pipeline import FramePrefetcher, ResultDrain


def test_prefetcher_decodes_each_frame_once_in_order():
    calls = []
    lock = threading.Lock()

    def decode(path):
        with lock:
            calls.append(path)
        time.sleep(0.002)
        return torch.full((1, 3, 4, 6), float(path))

    pf = FramePrefetcher(list(range(9)), "cpu", pad_fn=lambda t: torch.nn.functional.pad(t, (1, 1, 0, 0)), lookahead=3,
                         workers=3, decode=decode)
    for j in range(8):                       # the CLI's access pattern: (j, j+1) for consecutive pairs
        a, b = pf.get(j), pf.get(j + 1)
        assert a.shape == (1, 3, 4, 8) and float(a[0, 0, 0, 1]) == j and float(b[0, 0, 0, 1]) == j + 1
    pf.close()
    assert sorted(calls) == list(range(9)) and pf.decodes == 9          # every frame decoded exactly once
    assert len(pf.ready) <= 3                                           # sliding window, not the whole clip


def test_result_drain_runs_post_processing_off_thread_and_keeps_keys():
    drain = ResultDrain("cpu", depth=2)
    main_thread = threading.get_ident()
    seen = []

    def post(a, b):
        seen.append(threading.get_ident())
        return float(a.sum() + b.sum())

    for k in range(6):
        drain.submit(k, [torch.full((2, 2), float(k)), torch.ones(3)], post)
    out = drain.finish()
    assert out == {k: 4.0 * k + 3.0 for k in range(6)}
    assert all(t != main_thread for t in seen)


def test_result_drain_surfaces_errors():
    drain = ResultDrain("cpu")
    drain.submit(0, [torch.zeros(1)], lambda a: (_ for _ in ()).throw(ValueError("boom")))
    with pytest.raises(ValueError):
        drain.finish()
