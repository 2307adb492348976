"""Python view of the render entry points (include/rt_raytracer.h, include/rt_hip.h)."""
import ctypes as C
import threading
from dataclasses import dataclass

import numpy as np

from . import ctypes_abi as abi
from .native import lib as _lib, last_error
from .scene import HostScene, make_image


@dataclass
class Counters:
    paths: int = 0
    rays: int = 0
    node_visits: int = 0
    leaf_visits: int = 0
    shades: int = 0
    backgrounds: int = 0
    textured: int = 0

    @classmethod
    def from_struct(cls, s):
        return cls(*[int(getattr(s, f[0])) for f in s._fields_])

    def bytes_per_ray(self):
        """Algorithmic scene bytes per ray, SURVEY.md section 8d:
        192*N + 288*L + 112*H + 48*X + 12*M."""
        if self.rays == 0:
            return 0.0
        return (192.0 * self.node_visits + 288.0 * self.leaf_visits + 112.0 * self.shades
                + 48.0 * self.textured + 12.0 * self.backgrounds) / self.rays


def get_counters(lib=None) -> Counters:
    lib = lib or _lib
    c = abi.RT_Counters()
    if lib.rt_get_counters(C.byref(c)) != 0:
        raise RuntimeError(last_error(lib))
    return Counters.from_struct(c)


def render_frame(hs: HostScene, width, height, samples, max_bounces, seed=0x1234ABCD, want_linear=False,
                 want_accum=False, lib=None):
    """One frame on GPU 0 through rt_render_frame.  Returns dict(image u8 HxWx3, linear, accum, counters).
    lib: the library to render with (default: the product; tests pass raytracing_c_amd.diag for the wavefront pipeline)."""
    lib = lib or _lib
    lib.rt_set_seed(seed)
    out = np.zeros((height, width, 3), np.uint8)
    img, _keep = make_image(out)
    img.pixels.data = out.ctypes.data
    linear = np.zeros((height, width, 3), np.float32) if want_linear else None
    accum = np.zeros((height, width, 3), np.uint64) if want_accum else None
    rc = lib.rt_render_frame(C.byref(hs.scene), C.byref(img), samples, max_bounces,
                             linear.ctypes.data if want_linear else None,
                             accum.ctypes.data if want_accum else None)
    if rc != 0:
        raise RuntimeError("rt_render_frame failed: " + last_error(lib))
    return dict(image=out, linear=linear, accum=accum, counters=get_counters(lib))


def frame_begin(hs: HostScene, width, height, samples, max_bounces, seed=0x1234ABCD, lib=None):
    """rt_frame_begin (rt_hip.h): enqueue a frame, return (ticket, pixel array the frame will land in, keep-alive)."""
    lib = lib or _lib
    lib.rt_set_seed(seed)
    out = np.zeros((height, width, 3), np.uint8)
    img, keep = make_image(out)
    img.pixels.data = out.ctypes.data
    ticket = lib.rt_frame_begin(C.byref(hs.scene), C.byref(img), samples, max_bounces)
    if ticket < 0:
        raise RuntimeError("rt_frame_begin failed: " + last_error(lib))
    return ticket, out, (img, keep)


def frame_end(ticket, lib=None):
    """rt_frame_end: wait for the frame of `ticket`; its pixels are in the array frame_begin returned.  Returns its counters."""
    lib = lib or _lib
    if lib.rt_frame_end(ticket) != 0:
        raise RuntimeError("rt_frame_end failed: " + last_error(lib))
    return get_counters(lib)


def render_context(hs: HostScene, width, height, samples, max_bounces, n_threads=1, seed=0x1234ABCD, lib=None, fill=0):
    """The reference driver's protocol (driver.c:793-818): n_threads threads enter
    render_thread_proc on one Rendering_Context, the caller polls is_finished."""
    lib = lib or _lib
    lib.rt_set_seed(seed)
    out = np.full((height, width, 3), fill, np.uint8)
    ctx = abi.Rendering_Context()
    ctx.image.components = 3
    ctx.image.pixel_type = 0
    ctx.image.width = width
    ctx.image.stride = width
    ctx.image.height = height
    ctx.image.pixels.data = out.ctypes.data
    ctx.image.pixels.len = out.size
    ctx.scene = C.pointer(hs.scene)
    ctx.samples = samples
    ctx.max_bounces = max_bounces
    ctx.n_threads = n_threads
    ctx._current_chunk = 0
    threads = [threading.Thread(target=lib.render_thread_proc, args=(C.byref(ctx),)) for _ in range(n_threads)]
    for t in threads:
        t.start()
    lib.rendering_context_finish(C.byref(ctx))
    finished = bool(lib.rendering_context_is_finished(C.byref(ctx)))
    for t in threads:
        t.join()
    return dict(image=out, finished=finished, current_chunk=int(ctx._current_chunk), n_threads=int(ctx.n_threads))
