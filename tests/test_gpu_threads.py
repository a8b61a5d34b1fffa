"""Threading convention of the C ABI (the reference calls global_render from a non-main OS thread, gui.rs:182-185):
entry points may be called from any thread, one thread at a time per context; contexts are independent.  Four
threads, each with a context of its own, render different worlds at the same time; every image must equal the one
the main thread renders alone (same Philox streams; f32 atomics may move the last bit of a sum), and an error raised
in one thread must not show up in another's rt_last_error."""
import threading

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d

pytestmark = pytest.mark.gpu

WORLDS = ["complex_world", "sample_world", "cornell_box", "cornell_box_absorption"]


def test_contexts_on_concurrent_threads(gpu_ctx):
    p = params2d(96, 72, 24, 2, 10, seed=17)
    alone = {}
    for name in WORLDS:
        scene = rt.Scene2D(gpu_ctx, rt.World2D(name, seed=1))
        alone[name] = scene.global_render(p)[0]
        scene.close()
    got, errors = {}, []
    start = threading.Barrier(len(WORLDS))

    def work(name):
        try:
            ctx = rt.Context(0)
            scene = rt.Scene2D(ctx, rt.World2D(name, seed=1))
            start.wait(timeout=60)
            for _ in range(3):
                got[name] = scene.global_render(p)[0]
            if name == WORLDS[0]:
                # an error in this thread (an image size of zero) ...
                with pytest.raises(rt.RtError):
                    scene.global_render(params2d(0, 72, 1, 1, 10))
            else:
                # ... is not what another thread's successful call leaves behind
                scene.global_render(p)
            scene.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001 - reported below, in the main thread
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=work, args=(n,)) for n in WORLDS]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for name in WORLDS:
        assert np.allclose(got[name], alone[name], rtol=2e-4, atol=2e-5), name
