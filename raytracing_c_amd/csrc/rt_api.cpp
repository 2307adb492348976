// rt_api.cpp -- host side of librt_hip.so: the reference's render entry points
// (raytracer.h:51-56) and the device-control calls of include/rt_hip.h.
//
// Nothing in this file computes a pixel on the CPU: every entry point either
// drives the gfx950 kernels of rt_kernels.hip / rt_wavefront.hip or fails with rt_last_error().
//
// State is kept PER DEVICE (struct Device): HIP context, workspace, scene cache, launch timing.  Slot 0 is the
// process's primary device (rt_init); slots 1 .. N-1 exist when a frame behind render_thread_proc / render() is spread
// over N GPUs (RT_DEVICES, rt_set_devices).  The product library reads its configuration ONCE (config()); the RT_*
// experiment knobs of earlier rounds exist only in the diagnostic build (-DRT_DIAG_VARIANTS, librt_hip_diag.so).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_device.h"

// launchers in rt_kernels.hip
extern "C" {
int rt_launch_path_kernel(const RT_KParams *P, int n_waves, int variant, int smem_bytes, int wg_waves, hipStream_t stream);
int rt_launch_prepare(int n_tiles, uint32_t *tile_next, uint32_t *open_groups, unsigned long long *counters, uint32_t *work_head,
                      uint32_t *cost_cur, const uint32_t *cost_prev, uint32_t *order, hipStream_t stream);
int rt_launch_resolve(int width, int height, int samples, int chunks_x, const int32_t *local_chunks, int n_local_chunks,
                      const unsigned long long *accum, uint8_t *tiles, uint8_t *image, float *linear,
                      hipStream_t stream);
int rt_launch_untile(int width, int height, int chunks_x, int n_chunks, const int32_t *owner_slot,
                     const uint8_t *all_tiles, uint8_t *image, hipStream_t stream);
#ifdef RT_DIAG_VARIANTS      // unit-test kernels and the wavefront pipeline exist in the diagnostic library only
int rt_launch_test_math(int op, int n, const float *x, const float *y, float *out, hipStream_t stream);
int rt_launch_test_rcp_sweep(unsigned long long *counts, hipStream_t stream);
int rt_launch_test_srgb_sweep(unsigned long long *counts, hipStream_t stream);
int rt_launch_test_quantize_sweep(unsigned long long *counts, hipStream_t stream);
int rt_launch_test_trace(const RT_KParams *P, int n, const float *rays, float *out_t, int *out_tri, float *out_uv,
                         hipStream_t stream);
int rt_launch_test_trace_stream(const RT_KParams *P, int n, const float *rays, const float *pyr, int exit_lanes, int n_blocks,
                                int smem_bytes, float *out_t, int *out_tri, float *out_uv, unsigned long long *visits,
                                hipStream_t stream);
int rt_launch_test_texture(const RT_KParams *P, int tex, int n, const float *uv, float *out, hipStream_t stream);
#endif
int rt_launch_lightmap(const RT_KParams *P, const float *verts, int n_tris, int lw, int lh, int stride, int comp,
                       int samples, int *owner, uint8_t *pixels, hipStream_t stream);
int rt_launch_denoise(int width, int height, int src_stride, int src_comp, int dst_stride, int dst_comp,
                      const uint8_t *src, uint8_t *dst, hipStream_t stream);
int rt_launch_pack_texture(const uint8_t *raw, int width, int rows, int y0, int stride, int comp, uint32_t *out,
                           hipStream_t stream);
#ifdef RT_DIAG_VARIANTS
// rt_wavefront.hip
int rt_wf_launch_camera(const RT_KParams *P, int n_blocks, int geometry, int smem_bytes, hipStream_t stream);
int rt_wf_launch_trace(const RT_KParams *P, int n_blocks, int geometry, int smem_bytes, hipStream_t stream);
int rt_wf_launch_shade(const RT_KParams *P, int n_blocks, int first, hipStream_t stream);
#endif
}

// ---------------------------------------------------------------------------------
// errors

static std::mutex g_err_mutex;
static char       g_err[1024] = "";

static int rt_fail(const char *fmt, ...) {
  std::lock_guard<std::mutex> lock(g_err_mutex);
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  fprintf(stderr, "rt_hip: %s\n", g_err);
  return -1;
}

extern "C" char const *rt_last_error(void) { return g_err; }
extern "C" void rt_clear_error(void) {
  std::lock_guard<std::mutex> lock(g_err_mutex);
  g_err[0] = 0;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return rt_fail("%s failed: %s", #expr, hipGetErrorString(e_));      \
  } while (0)

// Temporary device buffer that is released on every exit path (HIP_TRY returns early).
struct DevBuf {
  void *p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T> T *as() const { return (T *)p; }
};

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------
// configuration: read once, never per launch
//
// Product library: RT_DEVICES (GPUs a frame behind render_thread_proc / render() is spread over, default 1) and
// RT_DEVICES_REHEARSE (=1: the N logical devices all map onto the primary GPU -- what a one-GPU box can run of the
// N-GPU path), overridable by rt_set_devices().  Nothing else in the environment changes what the library does.
// Diagnostic library (-DRT_DIAG_VARIANTS): the experiment knobs of earlier rounds (RT_KERNEL, RT_SCHED_THRESH, ...),
// read at every launch so that one process can A/B them (tools/exp_kernels.py, tests/test_gpu_diag.py).

#ifdef RT_DIAG_VARIANTS
static int knob_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}
static bool knob_is(const char *name, const char *value) {
  const char *e = getenv(name);
  return e && strcmp(e, value) == 0;
}
static bool knob_set(const char *name) { return getenv(name) != nullptr; }
#else
static inline int knob_int(const char *, int dflt) { return dflt; }
static inline bool knob_is(const char *, const char *) { return false; }
static inline bool knob_set(const char *) { return false; }
#endif

#define RT_MAX_DEVICES 16

struct Config {
  int  devices = 1;
  bool rehearse = false;
};
static std::mutex g_cfg_mutex;
static Config &config_locked() {
  static Config c = [] {
    Config c0;
    if (const char *e = getenv("RT_DEVICES")) {
      int v = atoi(e);
      if (v >= 1 && v <= RT_MAX_DEVICES) c0.devices = v;
    }
    if (const char *e = getenv("RT_DEVICES_REHEARSE")) c0.rehearse = atoi(e) != 0;
    return c0;
  }();
  return c;
}
static Config config() {
  std::lock_guard<std::mutex> lock(g_cfg_mutex);
  return config_locked();
}

static void remap_device_slots();
extern "C" int rt_set_devices(i32 n_devices, i32 rehearse) {
  if (n_devices < 1 || n_devices > RT_MAX_DEVICES) return rt_fail("rt_set_devices: %d outside [1, %d]", n_devices, RT_MAX_DEVICES);
  {
    std::lock_guard<std::mutex> lock(g_cfg_mutex);
    Config &c = config_locked();
    c.devices = n_devices;
    c.rehearse = rehearse != 0;
  }
  remap_device_slots();          // a slot that was mapped to another GPU under the old configuration starts over
  return 0;
}

#ifdef RT_DIAG_VARIANTS
// Diagnostic library only.  0 = tile-stream path kernel (the product's ONE kernel), 1 = wavefront pipeline (rt_wavefront.hip:
// same images and counters, measured slower on every BASELINE config -- profiles/r03_experiments.md -- kept for measurements)
static std::atomic<int>     g_pipeline{0};
static std::atomic<int64_t> g_wf_cap_records{(int64_t)96 << 20};

extern "C" int rt_set_pipeline(i32 pipeline) {
  if (pipeline != 0 && pipeline != 1) return rt_fail("rt_set_pipeline: %d is not 0 (tile stream) or 1 (wavefront)", pipeline);
  g_pipeline.store(pipeline);
  return 0;
}
extern "C" i32 rt_get_pipeline(void) { return g_pipeline.load(); }
extern "C" void rt_set_wavefront_capacity(i64 records) {
  if (records >= 1024) g_wf_cap_records.store(records);
}
#endif

// ---------------------------------------------------------------------------------
// per-device state

struct Partition;
struct RT_Device_Scene;

struct FrameTiming {          // the most recent frame through render_thread_proc / render / rt_render_frame
  float stamp_ms = 0, upload_ms = 0, enqueue_ms = 0, gpu_prep_ms = 0, gpu_path_ms = 0, gpu_resolve_ms = 0, gpu_copy_ms = 0,
        total_ms = 0, verify_ms = 0, gather_ms = 0;
  int   n_devices = 1, slowest_device = 0;
};

#ifndef RT_FRAME_LANES
#define RT_FRAME_LANES   2                       // frames in flight behind rt_frame_begin / rt_frame_end
#endif
#define RT_LAUNCH_STATES (1 + RT_FRAME_LANES)

struct Workspace {
  unsigned long long *accum = nullptr;
  size_t              accum_elems = 0;
  uint8_t            *image = nullptr;
  float              *linear = nullptr;
  size_t              image_pixels = 0;
  uint8_t            *tiles = nullptr;        // multi-device frames: this device's compact tiles
  size_t              tiles_bytes = 0;
  uint8_t            *all_tiles = nullptr;    // slot 0: the tiles of every device, rank-major
  size_t              all_tiles_bytes = 0;
  uint8_t            *tiles_host = nullptr;   // pinned: a device without peer access to slot 0 stages its tiles here
  size_t              tiles_host_bytes = 0;
  unsigned long long *counters_host = nullptr;   // pinned: ray counters of a multi-device frame, copied asynchronously
  // HIP event pairs around every path-kernel launch since the last timing reset
  std::vector<hipEvent_t> ev0, ev1;
  size_t              n_timed = 0;
  hipEvent_t          ev_frame[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // frame start, prep done, path done, resolve done, copy done
  unsigned long long *wave_times = nullptr;   // diagnostic kernel (RT_KERNEL=4) / RT_WAVE_TIMES
  int                 wave_times_n = 0;
};
#define RT_MAX_TIMED 256

struct DevPartition {
  const Partition        *host = nullptr;
  std::vector<int32_t *>  d_lists;            // device copies of the ranks' chunk lists, uploaded on first use
  int32_t                *d_owner_slot = nullptr;
};

// A frame in flight behind rt_frame_begin() / rt_frame_end() (slot 0): its own stream, accumulators and image buffer, and launch
// state 1 + lane of the device scene it renders from -- two frames overlap on the GPU, the second fills the CUs the first one's
// thinning bounce chains leave idle (profiles/r05_small_launch.md: a launch ends 0.6 - 1.0 ms after its last unit is handed out).
struct FrameLane {
  bool        busy = false;                   // begun, not yet ended
  bool        finished = false;               // rendered synchronously inside rt_frame_begin (a multi-device frame): nothing to wait for
  bool        ending = false;                 // a thread is inside rt_frame_end for this lane, waiting without the mutex
  int         rc = 0;
  hipStream_t stream = nullptr;
  Workspace   ws;
  Scene const *scene = nullptr;
  Image       image;                          // the caller's Image header (the pixels stay the caller's)
  RT_Render_Params p;
  Camera      camera;                         // scene->camera when the frame began
  RT_Device_Scene *d = nullptr;               // the copy the frame renders from; nullptr once that copy was dropped (free_device_scene waits first)
  uint64_t    fp = 0;                         // full fingerprint of that copy when the frame began
  bool        verify = false;
  FrameTiming timing;
  double      t_begin = 0.0;
};

struct Device {
  int        slot = 0, phys = 0;
  bool       ready = false;
  bool       peer_ok = true;                  // slots >= 1: direct copies into slot 0's memory are possible (xGMI peer access)
  hipStream_t mstream = nullptr;              // multi-device frames: this slot's own stream (slots rehearsed on ONE GPU overlap on it)
  int        num_cus = 0;
  std::mutex mutex;                           // serialises frames, the scene cache and the workspace of this device
  Workspace  ws;
  unsigned long long *last_counters = nullptr;   // counters of the most recent path-kernel launch
  std::unordered_map<const Scene *, RT_Device_Scene *> scene_cache;
  std::unordered_map<RT_Device_Scene *, Camera>        cameras;
  std::vector<DevPartition>                            parts;   // guarded by g_partition_mutex
  FrameTiming timing;
  FrameLane  lanes[RT_FRAME_LANES];           // slot 0 only
};

static Device g_devs[RT_MAX_DEVICES];
static int    g_primary = 0;                   // physical device of slot 0
static bool   g_primary_fixed = false;         // slot 0 has been initialised (rt_init can no longer move it)
static std::atomic<u32> g_seed{0x1234ABCDu};
static std::mutex g_multi_mutex;               // counters of the last multi-device frame
static RT_Counters g_multi_counters;
static bool        g_multi_counters_valid = false;

static Device &dev0() { return g_devs[0]; }

// Makes `D`'s GPU the calling thread's current HIP device for the guard's lifetime.
struct DeviceGuard {
  int  prev = -1;
  bool switched = false;
  explicit DeviceGuard(const Device &D) {
    if (hipGetDevice(&prev) == hipSuccess && prev != D.phys) switched = hipSetDevice(D.phys) == hipSuccess;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

static int ensure_device(Device &D) {            // D.mutex held (or single-threaded start-up)
  if (D.ready) return hipSetDevice(D.phys) == hipSuccess ? 0 : rt_fail("hipSetDevice(%d) failed", D.phys);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    return rt_fail("no HIP device available (%s); the render path has no CPU fallback",
                   e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  }
  if (D.slot == 0) {
    D.phys = g_primary;
    g_primary_fixed = true;
  } else {
    D.phys = config().rehearse ? g_primary : (g_primary + D.slot) % count;
  }
  if (D.phys >= count) return rt_fail("device %d requested but only %d present", D.phys, count);
  HIP_TRY(hipSetDevice(D.phys));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, D.phys));
  D.num_cus = prop.multiProcessorCount;
  if (D.slot != 0 && D.phys != g_primary) {
    // this device sends its tiles into device 0's buffer: direct xGMI copies when peer access can be enabled, staged ones otherwise
    int can = 0;
    D.peer_ok = false;
    if (hipDeviceCanAccessPeer(&can, D.phys, g_primary) == hipSuccess && can) {
      hipError_t pe = hipDeviceEnablePeerAccess(g_primary, 0);
      if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) D.peer_ok = true;
      if (pe != hipSuccess) (void)hipGetLastError();
    }
    // (refused: the tiles go through a pinned host buffer, render_frame_multi)
  }
  if (!D.mstream) HIP_TRY(hipStreamCreateWithFlags(&D.mstream, hipStreamNonBlocking));
  D.ready = true;
  return 0;
}

extern "C" int rt_init(int device) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (g_primary_fixed && device != g_primary) return rt_fail("rt_init: device already initialised as %d", g_primary);
  if (device < 0) return rt_fail("rt_init: device %d is invalid", device);
  g_primary = device;
  for (int i = 0; i < RT_MAX_DEVICES; i++) g_devs[i].slot = i;
  (void)config();                                 // the one read of the environment
  return ensure_device(D);
}

// GPUs a frame behind render_thread_proc / render() will be spread over on this machine right now
extern "C" i32 rt_device_count(void) {
  Config c = config();
  if (c.rehearse) return c.devices;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return 0;
  return c.devices < count ? c.devices : count;
}

extern "C" void rt_set_seed(u32 seed) { g_seed.store(seed); }
extern "C" u32  rt_get_seed(void) { return g_seed.load(); }

// ---------------------------------------------------------------------------------
// material tokens (rt_materials.h): recognised by address, not callable

extern "C" void disney_shader_proc(rawptr, Shader_Input const *, Shader_Output *output) {
  rt_fail("disney_shader_proc is a device material token and cannot be called on the host");
  if (output) output->terminate = true;
}

extern "C" void debug_shader_proc(rawptr, Shader_Input const *, Shader_Output *output) {
  rt_fail("debug_shader_proc is a device material token and cannot be called on the host");
  if (output) output->terminate = true;
}

extern "C" Color3 sample_background(Image const *, Vec3) {
  rt_fail("sample_background is a device background token and cannot be called on the host");
  Color3 c;
  c.x = c.y = c.z = 0.0f;
  return c;
}

// What rt_scene_upload compares Shader.proc / Background.proc with: this library's own exported tokens.  The diagnostic
// library can be told to recognise the PRODUCT library's tokens instead (rt_diag_set_tokens): a test process that has both
// libraries mapped builds its scenes once, with the product's addresses, and sends them through the unit-test entry points
// of the diagnostic build.
static Shader_Proc     g_tok_disney = disney_shader_proc;
static Shader_Proc     g_tok_debug = debug_shader_proc;
static Background_Proc g_tok_background = (Background_Proc)sample_background;
#ifdef RT_DIAG_VARIANTS
extern "C" void rt_diag_set_tokens(void *disney, void *debug, void *background) {
  if (disney) g_tok_disney = (Shader_Proc)disney;
  if (debug) g_tok_debug = (Shader_Proc)debug;
  if (background) g_tok_background = (Background_Proc)background;
}
#endif

// ---------------------------------------------------------------------------------
// scene residency

struct FpBlock {            // one block of a host scene's full fingerprint (scene_fingerprint_blocks)
  const void *begin;
  size_t      bytes;
  uint64_t    h;
};

// What ONE launch of the path kernel writes besides the accumulators: counters, work head, the tiles' unit counters, parked hits,
// the schedule feedback.  A device scene owns several (RT_Device_Scene::ls), allocated on first use.
struct LaunchState {
  unsigned long long *counters = nullptr;      // RT_N_COUNTERS
  uint32_t           *work_head = nullptr;
  uint32_t           *tile_next = nullptr;     // tile-stream kernel: chunks handed out per tile
  int32_t             tile_next_n = 0;
  uint32_t           *park = nullptr;          // tile-stream kernel: parked hits, [waves][18][128]
  int32_t             park_waves = 0;
  // schedule feedback: rays per 8x8 tile of the previous launch of the same frame shape -> visiting order of the next
  uint32_t    *cost[2] = {nullptr, nullptr};   // [cur] is written by the running launch, [cur^1] is last launch's
  uint32_t    *order = nullptr;
  int32_t      sched_tiles = 0, sched_cur = 0;
  bool         sched_valid = false;            // cost[cur^1] holds the costs of a launch with sched_key
  uint64_t     sched_key = 0;
};

struct RT_Device_Scene {
  Device      *dev = nullptr;
  float       *nodes = nullptr;
  float       *leaves = nullptr;
  float       *tris = nullptr;
  float       *mats = nullptr;
  RT_DTexture *textures = nullptr;
  uint32_t    *texels = nullptr;
  int32_t      depth = 0, last_row_offset = 0, bg_texture = -1, n_nodes = 0;
  int32_t      n_triangles = 0, n_materials = 0, n_textures = 0;
  int64_t      bytes = 0;
  float        max_edge = 0.0f;   // largest |component| of an edge b - a, c - a in the leaf tiles (NaN if one is NaN)
  bool         boxes_ordered = true;      // every child box of every node has min <= max on every axis (no NaN either)
  // what the host Scene looked like at upload: the per-frame stamp (scene_stamp) re-reads exactly this much of it
  uint64_t     stamp = 0;
  std::vector<const void *> mat_ptrs;     // distinct shader.data pointers, upload order
  std::vector<int32_t>      mat_first_tri;   // a triangle that uses mat_ptrs[k]
  // the FULL fingerprint of the host scene this copy was made from (scene_fingerprint_blocks), refreshed by rt_scene_touch:
  // every device slot checks a frame against ITS OWN copy's value
  std::vector<FpBlock>      fp_blocks;
  uint64_t                  full_fp = 0;
  // what rt_scene_touch() needs to patch the copy in place
  std::unordered_map<uint64_t, int> mat_map;            // (shader.data, kind) -> material id
  std::vector<const Image *>        tex_sources;        // Image of texture k (pool order; the background is one of them)
  std::vector<RT_DTexture>          tex_descs;          // its slot in the texel pool
  // launch state: RT_LAUNCH_STATES of them, so that launches of ONE device scene can be in flight on several streams at once
  // ([0]: the blocking entry points and rt_render_accumulate; [1 + k]: frame lane k of rt_frame_begin / rt_frame_end)
  LaunchState ls[RT_LAUNCH_STATES];
  // wavefront pipeline (rt_wavefront.hip): record queues between the camera / shade / trace kernels
  uint32_t           *wf_hit0 = nullptr, *wf_hit = nullptr, *wf_ray[2] = {nullptr, nullptr};
  uint32_t           *wf_cnt = nullptr;        // records per chunk: hit0 | hit | ray[0] | ray[1]
  uint32_t           *wf_ctl = nullptr;        // WF_N_CTL control words, one per 64-byte line
  uint32_t           *wf_ctl_host = nullptr;   // pinned copy the host reads after a pass
  int64_t             wf_soft0 = 0, wf_hard0 = 0, wf_ray_chunks = 0, wf_hit_chunks = 0;   // capacities in chunks
  int32_t             wf_waves = 0;            // waves the capacities were sized for
};

static void free_device_scene(RT_Device_Scene *d) {      // d->dev->mutex held, d's device current
  if (!d) return;
  if (d->dev)
    for (FrameLane &F : d->dev->lanes)        // a frame in flight renders from this copy: let it finish (its pixels are in the lane's buffer)
      if (F.busy && F.d == d) {
        if (F.stream) (void)hipStreamSynchronize(F.stream);
        F.d = nullptr;
      }
  (void)hipFree(d->nodes);
  (void)hipFree(d->leaves);
  (void)hipFree(d->tris);
  (void)hipFree(d->mats);
  (void)hipFree(d->textures);
  (void)hipFree(d->texels);
  for (LaunchState &L : d->ls) {
    if (d->dev && L.counters && d->dev->last_counters == L.counters) d->dev->last_counters = nullptr;
    (void)hipFree(L.counters);
    (void)hipFree(L.work_head);
    (void)hipFree(L.tile_next);
    (void)hipFree(L.park);
    (void)hipFree(L.cost[0]);
    (void)hipFree(L.cost[1]);
    (void)hipFree(L.order);
  }
  (void)hipFree(d->wf_hit0);
  (void)hipFree(d->wf_hit);
  (void)hipFree(d->wf_ray[0]);
  (void)hipFree(d->wf_ray[1]);
  (void)hipFree(d->wf_cnt);
  (void)hipFree(d->wf_ctl);
  if (d->wf_ctl_host) (void)hipHostFree(d->wf_ctl_host);
  delete d;
}

template <typename T>
static int upload(T **dst, const std::vector<T> &src, int64_t *bytes) {
  size_t n = src.size() * sizeof(T);
  if (n == 0) n = 16;
  HIP_TRY(hipMalloc((void **)dst, n));
  if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  *bytes += (int64_t)n;
  return 0;
}

// Textures are packed to RGBA8 words ON THE GPU: the host only records which Images are used (in first-use order)
// and where each one starts in the texel pool; upload_textures() copies the raw rows and runs rt_pack_texture_kernel.
struct TexturePool {
  std::unordered_map<const Image *, int> map;
  std::vector<RT_DTexture>               descs;
  std::vector<const Image *>             sources;
  size_t                                 texels = 0;
};

static int texture_index(Image const *img, TexturePool &pool) {
  if (!img) return -1;
  auto it = pool.map.find(img);
  if (it != pool.map.end()) return it->second;
  if (img->pixel_type != PT_u8 || img->components < 3 || img->width <= 0 || img->height <= 0 || !img->pixels.data ||
      img->stride < img->width ||
      img->pixels.len < (isize)img->stride * img->height * img->components) {     // the upload reads exactly that many bytes
    return -2;
  }
  RT_DTexture t;
  t.offset = (uint32_t)pool.texels;
  t.width = (int32_t)img->width;
  t.height = (int32_t)img->height;
#if RT_TEX_TILED
  t.stride = (int32_t)((img->width + 3) / 4);                                  // tiles per row (rt_device.h)
  pool.texels += (size_t)t.stride * (size_t)((img->height + 3) / 4) * 16;
#else
  t.stride = (int32_t)img->width;
  pool.texels += (size_t)img->width * img->height;
#endif
  int idx = (int)pool.descs.size();
  pool.descs.push_back(t);
  pool.sources.push_back(img);
  pool.map[img] = idx;
  return idx;
}

static int upload_textures(const TexturePool &pool, uint32_t **d_texels, int64_t *bytes) {
  size_t n = pool.texels ? pool.texels * 4 : 16;
  HIP_TRY(hipMalloc((void **)d_texels, n));
  *bytes += (int64_t)n;
  size_t max_raw = 0;
  for (const Image *img : pool.sources) {
    size_t raw = (size_t)img->stride * img->height * img->components;
    if (raw > max_raw) max_raw = raw;
  }
  if (max_raw == 0) return 0;
  DevBuf stage_buf;
  HIP_TRY(stage_buf.alloc(max_raw));
  uint8_t *stage = stage_buf.as<uint8_t>();
  int rc = 0;
  for (size_t k = 0; k < pool.sources.size() && rc == 0; k++) {
    const Image *img = pool.sources[k];
    size_t raw = (size_t)img->stride * img->height * img->components;
    rc = (int)hipMemcpy(stage, img->pixels.data, raw, hipMemcpyHostToDevice);
    if (rc == 0) rc = rt_launch_pack_texture(stage, (int)img->width, (int)img->height, 0, (int)img->stride, (int)img->components,
                                             *d_texels + pool.descs[k].offset, nullptr);
    if (rc == 0) rc = (int)hipDeviceSynchronize();        // the staging buffer is reused by the next texture
  }
  if (rc != 0) return rt_fail("texture upload failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}


// ---- stamps of a host Scene ---------------------------------------------------------------------------------
// render_thread_proc / render / lightmap_bake keep one device copy per Scene* and must notice when the host scene
// changed underneath it -- the reference reads the live Scene every frame.  Two levels:
//  * scene_stamp(), paid on EVERY frame (tens of microseconds): the dimensions and base pointers of the BVH and the
//    triangle block, the background proc and Image, every distinct material record (PBR_Shader_Data, 80 bytes) and the
//    descriptor (pointer, size, layout) of every Image a material references, IN FULL; of every block of geometry or texel
//    bytes a bounded sample (hash_sampled: blocks up to 4 KB in full, larger ones 8 runs of 512 bytes).  A rebuilt,
//    reloaded, regenerated or re-materialed scene, a changed material parameter, a swapped texture and any edit of a small
//    scene are noticed.
//  * What it can miss: an IN-PLACE edit of a few vertices or texels inside a large block (the helmet has 5 MB of geometry
//    and 50 MB of texels; hashing them, even one word in 61, cost 0.5 ms per frame in round 2).  A host that does that
//    calls rt_scene_invalidate(scene) afterwards; scene_init / scene_init_sah / scene_init_gpu /
//    scene_load_bytes do so themselves.  scene_fingerprint() -- geometry in full, texels of large images one word in 61 --
//    is what rt_scene_verify() compares for a host that wants the check anyway.
static inline uint64_t mix64(uint64_t h, uint64_t v) {
  h ^= v;
  h *= 0x9E3779B97F4A7C15ull;
  return h ^ (h >> 29);
}

static uint64_t hash_bytes(uint64_t seed, const void *data, size_t n, size_t stride_words = 1) {
  const unsigned char *b = (const unsigned char *)data;
  uint64_t h0 = seed ^ 0x243F6A8885A308D3ull, h1 = seed ^ 0x13198A2E03707344ull;
  uint64_t h2 = seed ^ 0xA4093822299F31D0ull, h3 = seed ^ 0x082EFA98EC4E6C89ull;
  size_t words = n / 8, i = 0;
  const size_t step = 4 * stride_words;
  for (; i + step <= words; i += step) {         // four independent chains: the multiplies pipeline
    uint64_t w0, w1, w2, w3;
    memcpy(&w0, b + 8 * i, 8);
    memcpy(&w1, b + 8 * (i + stride_words), 8);
    memcpy(&w2, b + 8 * (i + 2 * stride_words), 8);
    memcpy(&w3, b + 8 * (i + 3 * stride_words), 8);
    h0 = mix64(h0, w0); h1 = mix64(h1, w1); h2 = mix64(h2, w2); h3 = mix64(h3, w3);
  }
  uint64_t h = mix64(mix64(mix64(h0, h1), h2), h3);
  for (; i < words; i += stride_words) { uint64_t w; memcpy(&w, b + 8 * i, 8); h = mix64(h, w); }
  if (stride_words == 1)
    for (size_t k = words * 8; k < n; k++) h = mix64(h, b[k]);
  return mix64(h, (uint64_t)n);
}

static uint64_t hash_image(uint64_t h, Image const *img) {
  if (!img) return mix64(h, 0x1234u);
  int64_t desc[6] = {(int64_t)img->components, (int64_t)img->pixel_type, (int64_t)img->width, (int64_t)img->stride,
                     (int64_t)img->height, (int64_t)(uintptr_t)img->pixels.data};
  h = hash_bytes(h, desc, sizeof desc);
  if (img->pixels.data && img->pixel_type == PT_u8 && img->width > 0 && img->height > 0 && img->stride >= img->width &&
      img->components > 0) {
    size_t n = (size_t)img->stride * img->height * img->components;
    if (img->pixels.len >= (isize)n) h = hash_bytes(h, img->pixels.data, n, n > 65536 ? 61 : 1);    // small images in full
  }
  return h;
}


// The full fingerprint as a list of BLOCKS -- the dimensions, the node array, each of the nine coordinate arrays, the AoS
// records, every distinct material record, the texels of every image it references, the background image -- each with the
// host range it covers: rt_scene_touch() uses the ranges to tell the block the host says it wrote from blocks that changed
// without a word (those drop the copy instead of being absorbed into the new reference).
static void fp_image(std::vector<FpBlock> &out, Image const *img) {
  if (!img) { out.push_back({nullptr, 0, mix64(0x1234u, 0)}); return; }
  size_t n = 0;
  if (img->pixels.data && img->pixel_type == PT_u8 && img->width > 0 && img->height > 0 && img->stride >= img->width && img->components > 0) {
    n = (size_t)img->stride * img->height * img->components;
    if (img->pixels.len < (isize)n) n = 0;
  }
  out.push_back({n ? img->pixels.data : nullptr, n, hash_image(0x51ED27u, img)});
}

static void scene_fingerprint_blocks(Scene const *scene, std::vector<FpBlock> &out) {
  out.clear();
  const Triangles &T = scene->triangles;
  const uint64_t h0 = 0x452821E638D01377ull;
  int64_t dims[5] = {(int64_t)scene->bvh.depth, (int64_t)scene->bvh.last_row_offset, (int64_t)scene->bvh.nodes.len,
                     (int64_t)T.len, (int64_t)(uintptr_t)scene->background.proc};
  out.push_back({nullptr, 0, hash_bytes(h0, dims, sizeof dims)});
  if (scene->bvh.nodes.data && scene->bvh.nodes.len > 0) {
    const size_t n = (size_t)scene->bvh.nodes.len * sizeof(BVH_Node);
    out.push_back({scene->bvh.nodes.data, n, hash_bytes(h0, scene->bvh.nodes.data, n)});
  }
  if (T.len > 0 && T.x[0] && T.aos) {
    for (int k = 0; k < 3; k++) {                // the nine coordinate arrays (one block in scene_init, but not required to be)
      const float *arrs[3] = {T.x[k], T.y[k], T.z[k]};
      for (const float *a : arrs) out.push_back({a, (size_t)T.len * 4, hash_bytes(h0, a, (size_t)T.len * 4)});
    }
    out.push_back({T.aos, (size_t)T.len * sizeof(Triangle_AOS), hash_bytes(h0, T.aos, (size_t)T.len * sizeof(Triangle_AOS))});
    const void *last = nullptr;                  // distinct material records, in first-use order
    std::vector<const void *> seen;
    for (int i = 0; i < T.len; i++) {
      const Shader &sh = T.aos[i].shader;
      if (!sh.data || sh.data == last) continue;
      last = sh.data;
      bool dup = false;
      for (const void *q : seen) if (q == sh.data) { dup = true; break; }
      if (dup) continue;
      seen.push_back(sh.data);
      if (sh.proc == g_tok_disney || sh.proc == g_tok_debug) {
        const PBR_Shader_Data *m = (const PBR_Shader_Data *)sh.data;
        out.push_back({m, sizeof *m, hash_bytes(h0, m, sizeof *m)});
        fp_image(out, m->texture_albedo);
        fp_image(out, m->texture_normal);
        fp_image(out, m->texture_metal_roughness);
        fp_image(out, m->texture_emission);
      }
      if (seen.size() > 4096) break;             // pathological material counts: the pointers are in the AoS hash anyway
    }
  }
  if (scene->background.proc == g_tok_background) fp_image(out, (Image const *)scene->background.data);
}

static uint64_t fp_fold(const std::vector<FpBlock> &blocks) {
  uint64_t h = 0x452821E638D01377ull;
  for (const FpBlock &b : blocks) h = mix64(mix64(h, b.h), (uint64_t)b.bytes);
  return h;
}

static uint64_t scene_fingerprint(Scene const *scene) {
  std::vector<FpBlock> blocks;
  scene_fingerprint_blocks(scene, blocks);
  return fp_fold(blocks);
}


// A bounded look at a block of bytes: blocks up to 4 KB in full, larger ones as 8 runs of 512 bytes spread evenly over the
// block (runs, not single words: a run is 8 consecutive cache lines, a strided word per line would cost a miss each).
static uint64_t hash_sampled(uint64_t h, const void *data, size_t n) {
  if (!data || n == 0) return mix64(h, 0x5EEDu);
  if (n <= 4096) return hash_bytes(h, data, n);
  const unsigned char *b = (const unsigned char *)data;
  const size_t run = 512, runs = 8, span = n - run;
  for (size_t k = 0; k < runs; k++) {
    size_t off = (span * k / (runs - 1)) & ~(size_t)7;
    h = hash_bytes(h, b + off, run);
  }
  return mix64(h, (uint64_t)n);
}

static uint64_t hash_image_desc(uint64_t h, Image const *img) {
  if (!img) return mix64(h, 0x1234u);
  int64_t desc[6] = {(int64_t)img->components, (int64_t)img->pixel_type, (int64_t)img->width, (int64_t)img->stride,
                     (int64_t)img->height, (int64_t)(uintptr_t)img->pixels.data};
  h = hash_bytes(h, desc, sizeof desc);
  if (img->pixels.data && img->pixel_type == PT_u8 && img->width > 0 && img->height > 0 && img->stride >= img->width &&
      img->components > 0) {
    size_t n = (size_t)img->stride * img->height * img->components;
    if (img->pixels.len >= (isize)n) h = hash_sampled(h, img->pixels.data, n);
  }
  return h;
}

// `mat_ptrs` / `first_tri`: the distinct material records found at upload and one triangle that uses each.  A material
// pointer is only dereferenced while that triangle still points at it (a host that replaced its materials has freed
// the old records).  Returns 0 -- never a valid stamp -- when the scene no longer matches the lists.
static uint64_t scene_stamp(Scene const *scene, const std::vector<const void *> &mat_ptrs, const std::vector<int32_t> &first_tri) {
  const Triangles &T = scene->triangles;
  int64_t head[9] = {(int64_t)scene->bvh.depth, (int64_t)scene->bvh.last_row_offset, (int64_t)scene->bvh.nodes.len, (int64_t)T.len,
                     (int64_t)(uintptr_t)scene->background.proc, (int64_t)(uintptr_t)scene->background.data,
                     (int64_t)(uintptr_t)scene->bvh.nodes.data, (int64_t)(uintptr_t)T.x[0], (int64_t)(uintptr_t)T.aos};
  uint64_t h = hash_bytes(0x452821E638D01377ull, head, sizeof head);
  if (scene->bvh.nodes.data && scene->bvh.nodes.len > 0) h = hash_sampled(h, scene->bvh.nodes.data, (size_t)scene->bvh.nodes.len * sizeof(BVH_Node));
  if (T.len > 0 && T.x[0] && T.aos) {
    for (int k = 0; k < 3; k++) {
      h = hash_sampled(h, T.x[k], (size_t)T.len * 4);
      h = hash_sampled(h, T.y[k], (size_t)T.len * 4);
      h = hash_sampled(h, T.z[k], (size_t)T.len * 4);
    }
    h = hash_sampled(h, T.aos, (size_t)T.len * sizeof(Triangle_AOS));
  }
  if (T.aos)
    for (size_t k = 0; k < mat_ptrs.size(); k++) {
      const int32_t i = first_tri[k];
      if (i < 0 || i >= T.len || T.aos[i].shader.data != mat_ptrs[k]) return 0;
      const PBR_Shader_Data *m = (const PBR_Shader_Data *)mat_ptrs[k];
      h = hash_bytes(h, m, sizeof *m);
      h = mix64(h, (uint64_t)(uintptr_t)T.aos[i].shader.proc);
      h = hash_image_desc(h, m->texture_albedo);
      h = hash_image_desc(h, m->texture_normal);
      h = hash_image_desc(h, m->texture_metal_roughness);
      h = hash_image_desc(h, m->texture_emission);
    }
  if (scene->background.proc == g_tok_background) h = hash_image_desc(h, (Image const *)scene->background.data);
  return h | 1ull;
}

static float int_bits(int32_t i) {
  float f;
  memcpy(&f, &i, 4);
  return f;
}

// one leaf tile: 9 rows x 8: vertex a, then the edges b - a and c - a (raytracer.c:115-122 computes them per visit; the fp32
// subtraction done here gives the same bits)
static void build_leaf_tile(const Triangles &T, int g, float *l) {
  for (int k = 0; k < 8; k++) {
    int i = g * 8 + k;
    volatile float e;     // keep every difference a plain IEEE fp32 subtraction
    l[0 * 8 + k] = T.x[0][i];
    e = T.x[1][i] - T.x[0][i]; l[1 * 8 + k] = e;
    e = T.x[2][i] - T.x[0][i]; l[2 * 8 + k] = e;
    l[3 * 8 + k] = T.y[0][i];
    e = T.y[1][i] - T.y[0][i]; l[4 * 8 + k] = e;
    e = T.y[2][i] - T.y[0][i]; l[5 * 8 + k] = e;
    l[6 * 8 + k] = T.z[0][i];
    e = T.z[1][i] - T.z[0][i]; l[7 * 8 + k] = e;
    e = T.z[2][i] - T.z[0][i]; l[8 * 8 + k] = e;
  }
}
static void leaf_tile_max_edge(const float *l, float &max_edge) {
  for (int row : {1, 2, 4, 5, 7, 8})
    for (int k = 0; k < 8; k++) {
      float m = fabsf(l[row * 8 + k]);
      if (!(m <= max_edge)) max_edge = m;          // a NaN sticks (every later comparison is false as well)
    }
}
// the 28-float shading record of triangle i (rt_device.h) with material id `mat`
static void build_tri_record(const Triangle_AOS &a, int mat, float *r) {
  r[0] = a.normal.x;    r[1] = a.normal.y;    r[2] = a.normal.z;    r[3] = int_bits(mat < 0 ? 0 : mat);
  r[4] = a.normal_a.x;  r[5] = a.normal_a.y;  r[6] = a.normal_a.z;  r[7] = a.tex_coords_a.x;
  r[8] = a.normal_b.x;  r[9] = a.normal_b.y;  r[10] = a.normal_b.z; r[11] = a.tex_coords_a.y;
  r[12] = a.normal_c.x; r[13] = a.normal_c.y; r[14] = a.normal_c.z; r[15] = a.tex_coords_b.x;
  r[16] = a.tangent.x;  r[17] = a.tangent.y;  r[18] = a.tangent.z;  r[19] = a.tex_coords_b.y;
  r[20] = a.bitangent.x; r[21] = a.bitangent.y; r[22] = a.bitangent.z; r[23] = a.tex_coords_c.x;
  r[24] = a.tex_coords_c.y; r[25] = 0.0f; r[26] = 0.0f; r[27] = 0.0f;
}
static void build_material_row(const PBR_Shader_Data *d, int ta, int tn, int tm, int te, int kind,
                               const std::vector<RT_DTexture> &descs, float *m) {
  const float row[20] = {d->base_color.x, d->base_color.y, d->base_color.z, d->roughness,
                         d->emission.x, d->emission.y, d->emission.z, d->metalness,
                         d->normal_map_strength, d->sheen, d->sheen_tint, d->anisotropic_strength,
                         int_bits(ta), int_bits(tn), int_bits(tm), int_bits(te),
                         int_bits(kind), 0.0f, 0.0f, 0.0f};
  memcpy(m, row, sizeof row);
  const int tex[4] = {ta, tn, tm, te};
  for (int k = 0; k < 4; k++) {
    float *q = m + 20 + 4 * k;
    if (tex[k] >= 0 && (size_t)tex[k] < descs.size()) {
      const RT_DTexture &t = descs[(size_t)tex[k]];
      q[0] = int_bits((int32_t)t.offset); q[1] = int_bits(t.width); q[2] = int_bits(t.height); q[3] = int_bits(t.stride);
    } else {
      q[0] = q[1] = q[2] = q[3] = 0.0f;
    }
  }
}

static RT_Device_Scene *upload_scene_locked(Device &D, Scene const *scene) {
  if (ensure_device(D) != 0) return nullptr;
  if (!scene) { rt_fail("rt_scene_upload: scene is NULL"); return nullptr; }
  const Triangles &T = scene->triangles;
  if (T.len <= 0 || T.len % 8 != 0 || !T.x[0] || !T.aos) {
    rt_fail("rt_scene_upload: triangle block is empty or not a multiple of 8 (len=%d)", (int)T.len);
    return nullptr;
  }
  if (scene->bvh.depth < 0 || scene->bvh.depth > RT_MAX_DEPTH) {
    rt_fail("rt_scene_upload: bvh.depth %ld outside [0, %d]", (long)scene->bvh.depth, RT_MAX_DEPTH);
    return nullptr;
  }
  if (scene->bvh.depth > 0) {
    if (scene->bvh.nodes.len < bvh_n_internal_nodes(scene->bvh.depth) || !scene->bvh.nodes.data) {
      rt_fail("rt_scene_upload: %ld nodes for depth %ld", (long)scene->bvh.nodes.len, (long)scene->bvh.depth);
      return nullptr;
    }
    if ((isize)T.len < bvh_n_leaf_nodes(scene->bvh.depth) * 8) {
      rt_fail("rt_scene_upload: %d triangle slots for depth %ld", (int)T.len, (long)scene->bvh.depth);
      return nullptr;
    }
  }
  if (scene->background.proc != g_tok_background || !scene->background.data) {
    rt_fail("rt_scene_upload: background.proc is not the exported sample_background token "
            "(host callbacks cannot run on the GPU)");
    return nullptr;
  }

  TexturePool pool;
  std::unordered_map<uint64_t, int>      mat_map;     // (data ptr, kind) -> id
  std::vector<float>                     mats;
  std::vector<const void *>              mat_ptrs;
  std::vector<int32_t>                   mat_first_tri;

  // triangles + materials
  const int n = T.len;
  std::vector<float> tris((size_t)n * 28, 0.0f);
  for (int i = 0; i < n; i++) {
    const Triangle_AOS &a = T.aos[i];
    int mat = 0;
    if (a.shader.proc == nullptr) {
      mat = -1;   // unpopulated slot: never hit (all-zero triangle)
    } else {
      int kind;
      if (a.shader.proc == g_tok_disney) kind = RT_MAT_DISNEY;
      else if (a.shader.proc == g_tok_debug) kind = RT_MAT_DEBUG;
      else {
        rt_fail("rt_scene_upload: triangle %d uses a shader proc that is not an exported device material "
                "(disney_shader_proc / debug_shader_proc)", i);
        return nullptr;
      }
      if (!a.shader.data) { rt_fail("rt_scene_upload: triangle %d has shader.data == NULL", i); return nullptr; }
      uint64_t key = (uint64_t)(uintptr_t)a.shader.data * 2u + (uint64_t)kind;
      auto it = mat_map.find(key);
      if (it != mat_map.end()) {
        mat = it->second;
      } else {
        const PBR_Shader_Data *d = (const PBR_Shader_Data *)a.shader.data;
        int ta = texture_index(d->texture_albedo, pool);
        int tn = texture_index(d->texture_normal, pool);
        int tm = texture_index(d->texture_metal_roughness, pool);
        int te = texture_index(d->texture_emission, pool);
        if (ta == -2 || tn == -2 || tm == -2 || te == -2) {
          rt_fail("rt_scene_upload: material of triangle %d references an unusable Image (need PT_u8, >= 3 components, pixels.len >= stride*height*components)", i);
          return nullptr;
        }
        mat = (int)(mats.size() / RT_MAT_FLOATS);
        float m[RT_MAT_FLOATS];
        build_material_row(d, ta, tn, tm, te, kind, pool.descs, m);
        mats.insert(mats.end(), m, m + RT_MAT_FLOATS);
        mat_map[key] = mat;
        mat_ptrs.push_back(a.shader.data);
        mat_first_tri.push_back(i);
      }
    }
    build_tri_record(a, mat, &tris[(size_t)i * 28]);
  }
  if (mats.empty()) mats.resize(RT_MAT_FLOATS, 0.0f);

  // leaf tiles (build_leaf_tile)
  const int n_groups = n / 8;
  std::vector<float> leaves((size_t)n_groups * 72);
  float max_edge = 0.0f;
  for (int g = 0; g < n_groups; g++) {
    build_leaf_tile(T, g, &leaves[(size_t)g * 72]);
    leaf_tile_max_edge(&leaves[(size_t)g * 72], max_edge);
  }

  std::vector<float> nodes;
  bool boxes_ordered = true;
  if (scene->bvh.depth > 0) {
    const float *src = (const float *)scene->bvh.nodes.data;
    nodes.assign(src, src + (size_t)scene->bvh.nodes.len * 48);
    // the LDS node blocks pick the near / far plane of a slab by the sign of the ray direction: valid for min <= max
    // (what scene_init builds; a Scene from elsewhere is checked, and traverses through the min / max form otherwise)
    for (size_t nd = 0; nd < (size_t)scene->bvh.nodes.len && boxes_ordered; nd++)
      for (int k = 0; k < 24; k++)
        if (!(nodes[nd * 48 + k] <= nodes[nd * 48 + 24 + k])) { boxes_ordered = false; break; }
  }

  int bg = texture_index((Image const *)scene->background.data, pool);
  if (bg < 0) { rt_fail("rt_scene_upload: background Image is unusable (need PT_u8, >= 3 components, pixels.len >= stride*height*components)"); return nullptr; }

  RT_Device_Scene *d = new RT_Device_Scene();
  d->dev = &D;
  if (hipMalloc((void **)&d->ls[0].counters, RT_N_COUNTERS * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc((void **)&d->ls[0].work_head, 64) != hipSuccess) {
    rt_fail("rt_scene_upload: out of device memory");
    free_device_scene(d);
    return nullptr;
  }
  if (upload(&d->nodes, nodes, &d->bytes) || upload(&d->leaves, leaves, &d->bytes) ||
      upload(&d->tris, tris, &d->bytes) || upload(&d->mats, mats, &d->bytes) ||
      upload(&d->textures, pool.descs, &d->bytes) || upload_textures(pool, &d->texels, &d->bytes)) {
    free_device_scene(d);
    return nullptr;
  }
  d->depth = (int32_t)scene->bvh.depth;
  d->last_row_offset = (int32_t)scene->bvh.last_row_offset;
  d->bg_texture = bg;
  d->n_nodes = (int32_t)scene->bvh.nodes.len;
  d->n_triangles = n;
  d->max_edge = max_edge;
  d->boxes_ordered = boxes_ordered;
  d->n_materials = (int32_t)(mats.size() / RT_MAT_FLOATS);
  d->n_textures = (int32_t)pool.descs.size();
  d->mat_ptrs = mat_ptrs;
  d->mat_first_tri = mat_first_tri;
  d->mat_map = mat_map;
  d->tex_sources = pool.sources;
  d->tex_descs = pool.descs;
  d->stamp = scene_stamp(scene, d->mat_ptrs, d->mat_first_tri);
  return d;
}

// rt_scene_upload / release / set_camera work on the primary device (one process per GPU under torch.distributed)
extern "C" RT_Device_Scene *rt_scene_upload(Scene const *scene) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  RT_Device_Scene *d = upload_scene_locked(D, scene);
  if (d) D.cameras[d] = scene->camera;
  return d;
}

extern "C" void rt_scene_release(RT_Device_Scene *dscene) {
  if (!dscene) return;
  Device &D = dscene->dev ? *dscene->dev : dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  DeviceGuard guard(D);
  for (auto it = D.scene_cache.begin(); it != D.scene_cache.end(); ++it) {
    if (it->second == dscene) { D.scene_cache.erase(it); break; }
  }
  D.cameras.erase(dscene);
  free_device_scene(dscene);
}

static void forget_static(Scene const *scene);
extern "C" void rt_scene_invalidate(Scene const *scene) {
  forget_static(scene);             // (a Scene rebuilt or freed and allocated again at this address starts checked, like a new one)
  for (int i = 0; i < RT_MAX_DEVICES; i++) {
    Device &D = g_devs[i];
    std::lock_guard<std::mutex> lock(D.mutex);
    auto it = D.scene_cache.find(scene);
    if (it != D.scene_cache.end()) {
      DeviceGuard guard(D);
      free_device_scene(it->second);
      D.scene_cache.erase(it);
    }
  }
}

extern "C" i64 rt_scene_device_bytes(RT_Device_Scene const *dscene) { return dscene ? dscene->bytes : 0; }

// ---- in-place edits of a resident scene ------------------------------------------------------------------------------------
// The reference reads the live Scene every frame (raytracer.c:596-612).  Here a frame renders from the device copy, and
// three things keep the two equal:
//  1. the sampled stamp (scene_stamp) before the frame is enqueued: rebuilt / reloaded / re-materialed scenes, microseconds;
//  2. the FULL check (scene_fingerprint: every byte of the BVH, the coordinate arrays, the AoS records and the materials, the
//     texels of large images one word in 61) computed by the calling thread WHILE the GPU renders the frame; if it differs
//     from the one taken at upload the frame is thrown away, the scene uploaded again and the frame rendered again -- a
//     frame on an unchanged scene pays nothing but host time hidden behind its own kernel.  rt_scene_set_static(scene, 1)
//     turns this off for a host that promises not to edit in place (or that calls rt_scene_touch / rt_scene_invalidate);
//  3. rt_scene_touch(scene, begin, bytes): "I wrote these bytes" -- the one block they belong to is patched on every device
//     (a few texture rows, a few leaf tiles ...) instead of 80 MB being uploaded again; this is also what catches a
//     single-texel edit that the 1-in-61 sampling of (2) can miss.
static std::mutex g_static_mutex;
static std::unordered_map<const Scene *, bool> g_static_scenes;
extern "C" void rt_scene_set_static(Scene const *scene, i32 is_static) {
  std::lock_guard<std::mutex> lock(g_static_mutex);
  if (is_static) g_static_scenes[scene] = true;
  else g_static_scenes.erase(scene);
}
static void forget_static(Scene const *scene) {
  std::lock_guard<std::mutex> lock(g_static_mutex);
  g_static_scenes.erase(scene);
}
static bool scene_is_static(Scene const *scene) {
  std::lock_guard<std::mutex> lock(g_static_mutex);
  return g_static_scenes.count(scene) != 0;
}

static bool range_in(const void *begin, size_t bytes, const void *block, size_t block_bytes, size_t *off) {
  const uintptr_t b = (uintptr_t)begin, k = (uintptr_t)block;
  if (!block || b < k || b + bytes > k + block_bytes) return false;
  *off = (size_t)(b - k);
  return true;
}

// Patches the device copy `d` (device D current, D.mutex held) for host bytes [begin, begin + bytes).  1 = patched,
// 0 = the range is not something that can be patched (the caller drops the copy), -1 = HIP error.
static int touch_device_scene(RT_Device_Scene *d, Scene const *scene, const void *begin, size_t bytes) {
  const Triangles &T = scene->triangles;
  const int n = (int)T.len;
  size_t off = 0;
  if ((int32_t)scene->bvh.nodes.len != d->n_nodes || n != d->n_triangles || (int32_t)scene->bvh.depth != d->depth) return 0;
  // BVH nodes: stored as they are
  if (d->n_nodes > 0 && range_in(begin, bytes, scene->bvh.nodes.data, (size_t)d->n_nodes * sizeof(BVH_Node), &off)) {
    const size_t n0 = off / sizeof(BVH_Node), n1 = (off + bytes + sizeof(BVH_Node) - 1) / sizeof(BVH_Node);
    const float *src = (const float *)scene->bvh.nodes.data;
    for (size_t nd = n0; nd < n1; nd++)
      for (int k = 0; k < 24; k++)
        if (!(src[nd * 48 + k] <= src[nd * 48 + 24 + k])) d->boxes_ordered = false;
    HIP_TRY(hipMemcpy(d->nodes + n0 * 48, src + n0 * 48, (n1 - n0) * sizeof(BVH_Node), hipMemcpyHostToDevice));
    return 1;
  }
  // coordinate arrays: the leaf tiles of the touched triangles
  for (int k = 0; k < 3; k++) {
    const float *arrs[3] = {T.x[k], T.y[k], T.z[k]};
    for (const float *arr : arrs) {
      if (!range_in(begin, bytes, arr, (size_t)n * 4, &off)) continue;
      const int g0 = (int)(off / 4) / 8, g1 = (int)((off + bytes + 3) / 4 + 7) / 8;
      std::vector<float> tiles((size_t)(g1 - g0) * 72);
      for (int g = g0; g < g1; g++) {
        build_leaf_tile(T, g, &tiles[(size_t)(g - g0) * 72]);
        leaf_tile_max_edge(&tiles[(size_t)(g - g0) * 72], d->max_edge);        // (the bound can only grow: conservative)
      }
      HIP_TRY(hipMemcpy(d->leaves + (size_t)g0 * 72, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
      return 1;
    }
  }
  // AoS records: the shading records of the touched triangles (their materials must be known ones)
  if (range_in(begin, bytes, T.aos, (size_t)n * sizeof(Triangle_AOS), &off)) {
    const int i0 = (int)(off / sizeof(Triangle_AOS)), i1 = (int)((off + bytes + sizeof(Triangle_AOS) - 1) / sizeof(Triangle_AOS));
    std::vector<float> recs((size_t)(i1 - i0) * 28);
    for (int i = i0; i < i1; i++) {
      const Triangle_AOS &a = T.aos[i];
      int mat = -1;
      if (a.shader.proc != nullptr) {
        int kind = a.shader.proc == g_tok_disney ? RT_MAT_DISNEY : a.shader.proc == g_tok_debug ? RT_MAT_DEBUG : -1;
        if (kind < 0 || !a.shader.data) return 0;
        auto it = d->mat_map.find((uint64_t)(uintptr_t)a.shader.data * 2u + (uint64_t)kind);
        if (it == d->mat_map.end()) return 0;                          // a material the copy does not have: upload again
        mat = it->second;
      }
      build_tri_record(a, mat, &recs[(size_t)(i - i0) * 28]);
    }
    HIP_TRY(hipMemcpy(d->tris + (size_t)i0 * 28, recs.data(), recs.size() * 4, hipMemcpyHostToDevice));
    return 1;
  }
  // a material record (its texture pointers must still be Images of the pool)
  for (size_t k = 0; k < d->mat_ptrs.size(); k++) {
    if (!range_in(begin, bytes, d->mat_ptrs[k], sizeof(PBR_Shader_Data), &off)) continue;
    const PBR_Shader_Data *m = (const PBR_Shader_Data *)d->mat_ptrs[k];
    auto tex = [&](Image const *img) -> int {
      if (!img) return -1;
      for (size_t t = 0; t < d->tex_sources.size(); t++) if (d->tex_sources[t] == img) return (int)t;
      return -2;
    };
    const int ta = tex(m->texture_albedo), tn = tex(m->texture_normal), tm = tex(m->texture_metal_roughness), te = tex(m->texture_emission);
    if (ta == -2 || tn == -2 || tm == -2 || te == -2) return 0;
    bool any = false;
    for (auto &kv : d->mat_map) {                                        // the record may serve both kinds
      if ((const void *)(uintptr_t)(kv.first / 2u) != d->mat_ptrs[k]) continue;
      float row[RT_MAT_FLOATS];
      build_material_row(m, ta, tn, tm, te, (int)(kv.first & 1u), d->tex_descs, row);
      HIP_TRY(hipMemcpy(d->mats + (size_t)kv.second * RT_MAT_FLOATS, row, sizeof row, hipMemcpyHostToDevice));
      any = true;
    }
    return any ? 1 : 0;
  }
  // texels of a texture (or of the background): the touched rows are packed again
  for (size_t t = 0; t < d->tex_sources.size(); t++) {
    const Image *img = d->tex_sources[t];
    const RT_DTexture &desc = d->tex_descs[t];
    if (img->width != desc.width || img->height != desc.height || img->components < 3 || img->pixel_type != PT_u8) continue;
    const size_t row_bytes = (size_t)img->stride * img->components;
    if (!range_in(begin, bytes, img->pixels.data, row_bytes * img->height, &off)) continue;
    const size_t r0 = off / row_bytes, r1 = (off + bytes + row_bytes - 1) / row_bytes;
    DevBuf stage;
    HIP_TRY(stage.alloc((r1 - r0) * row_bytes));
    HIP_TRY(hipMemcpy(stage.p, (const unsigned char *)img->pixels.data + r0 * row_bytes, (r1 - r0) * row_bytes, hipMemcpyHostToDevice));
    uint32_t *tex_base = d->texels + desc.offset;
#if !RT_TEX_TILED
    tex_base += r0 * (size_t)desc.width;           // (row-major: the kernel's row 0 is row r0 of the texture)
#endif
    int rc = rt_launch_pack_texture(stage.as<uint8_t>(), (int)img->width, (int)(r1 - r0), RT_TEX_TILED ? (int)r0 : 0, (int)img->stride,
                                    (int)img->components, tex_base, nullptr);
    if (rc == 0) rc = (int)hipDeviceSynchronize();
    if (rc != 0) return rt_fail("rt_scene_touch: texture rows: %s", hipGetErrorString((hipError_t)rc));
    return 1;
  }
  return 0;
}

// Does the reported range [begin, begin + bytes) touch block `b`?
static bool fp_block_holds(const FpBlock &b, const void *begin, size_t bytes) {
  const uintptr_t r0 = (uintptr_t)begin, r1 = r0 + bytes, b0 = (uintptr_t)b.begin, b1 = b0 + b.bytes;
  return b.begin && r0 < b1 && b0 < r1;
}

// 0 = every resident copy was patched in place, 1 = copies were dropped (the next frame uploads), -1 = error
extern "C" int rt_scene_touch(Scene const *scene, void const *begin, size_t bytes) {
  if (!scene || !begin || bytes == 0) return rt_fail("rt_scene_touch: NULL scene or empty range");
  // The host scene as it is NOW, block by block.  A copy may be patched only if the blocks that differ from what it was made
  // from are the one(s) the caller says it wrote: anything else is an edit nobody reported, and taking the new fingerprint
  // as the reference would hide it from every later check -- such a copy is dropped instead.
  std::vector<FpBlock> now;
  scene_fingerprint_blocks(scene, now);
  int dropped = 0;
  for (int i = 0; i < RT_MAX_DEVICES; i++) {
    Device &D = g_devs[i];
    std::lock_guard<std::mutex> lock(D.mutex);
    auto it = D.scene_cache.find(scene);
    if (it == D.scene_cache.end()) continue;
    DeviceGuard guard(D);
    RT_Device_Scene *d = it->second;
    bool unreported = now.size() != d->fp_blocks.size();
    for (size_t k = 0; !unreported && k < now.size(); k++) {
      const FpBlock &a = now[k], &b = d->fp_blocks[k];
      if (a.begin != b.begin || a.bytes != b.bytes) unreported = true;
      else if (a.h != b.h && !fp_block_holds(a, begin, bytes)) unreported = true;
    }
    int rc = unreported ? 0 : touch_device_scene(d, scene, begin, (size_t)bytes);
    if (rc == 1) {
      d->stamp = scene_stamp(scene, d->mat_ptrs, d->mat_first_tri);
      d->fp_blocks = now;
      d->full_fp = fp_fold(now);
    } else {
      free_device_scene(d);
      D.scene_cache.erase(it);
      dropped += 1;
      if (rc < 0) return -1;
    }
  }
  return dropped ? 1 : 0;
}

// Full content check of the cached device copies of `scene`: 1 = the host scene still equals what every copy was made from
// (geometry and material bytes in full, texels of large images sampled), 0 = it changed (the copies that differ are dropped
// on EVERY device, the next frame uploads again), -1 = no cached copy on the primary device.  The per-frame check is
// scene_stamp() above.
static int drop_stale_copies(Scene const *scene, uint64_t now, int first_slot) {      // returns how many copies were dropped
  int dropped = 0;
  for (int i = first_slot; i < RT_MAX_DEVICES; i++) {
    Device &D = g_devs[i];
    std::lock_guard<std::mutex> lock(D.mutex);
    auto it = D.scene_cache.find(scene);
    if (it == D.scene_cache.end() || it->second->full_fp == now) continue;
    DeviceGuard guard(D);
    (void)hipDeviceSynchronize();
    free_device_scene(it->second);
    D.scene_cache.erase(it);
    dropped += 1;
  }
  return dropped;
}
extern "C" int rt_scene_verify(Scene const *scene) {
  if (!scene) return -1;
  {
    Device &D = dev0();
    std::lock_guard<std::mutex> lock(D.mutex);
    if (D.scene_cache.find(scene) == D.scene_cache.end()) return -1;
  }
  const uint64_t now = scene_fingerprint(scene);
  return drop_stale_copies(scene, now, 0) == 0 ? 1 : 0;
}

static void scene_only_kparams(RT_KParams *K, RT_Device_Scene *d) {
  memset(K, 0, sizeof *K);
  K->nodes = d->nodes;
  K->leaves = d->leaves;
  K->tris = d->tris;
  K->mats = d->mats;
  K->textures = d->textures;
  K->texels = d->texels;
  K->depth = d->depth;
  K->last_row_offset = d->last_row_offset;
  K->bg_texture = d->bg_texture;
  K->n_nodes = d->n_nodes;
}

static RT_Device_Scene *cached_scene_locked(Device &D, Scene const *scene, float *stamp_ms, float *upload_ms) {
  if (!scene) { rt_fail("render: scene is NULL"); return nullptr; }
  const double t0 = now_ms();
  auto it = D.scene_cache.find(scene);
  if (it != D.scene_cache.end()) {
    RT_Device_Scene *d = it->second;
    const bool same = d->stamp == scene_stamp(scene, d->mat_ptrs, d->mat_first_tri);
    if (stamp_ms) *stamp_ms = (float)(now_ms() - t0);
    if (same) return d;
    free_device_scene(d);                        // the host scene changed since the upload
    D.scene_cache.erase(it);
  }
  const double t1 = now_ms();
  RT_Device_Scene *d = upload_scene_locked(D, scene);
  if (d) {
    D.scene_cache[scene] = d;
    scene_fingerprint_blocks(scene, d->fp_blocks);      // (every slot keeps its own: a slot can be uploaded long after slot 0)
    d->full_fp = fp_fold(d->fp_blocks);
  }
  if (upload_ms) *upload_ms = (float)(now_ms() - t1);
  return d;
}

// ---------------------------------------------------------------------------------
// rendering

extern "C" i32 rt_chunk_count(i32 width, i32 height) {
  if (width <= 0 || height <= 0) return 0;
  return ((width + RT_CHUNK_SIZE - 1) / RT_CHUNK_SIZE) * ((height + RT_CHUNK_SIZE - 1) / RT_CHUNK_SIZE);
}

// ---- image partition across ranks -------------------------------------------------------------------
// Chunk (cx, cy) of the 32x32 grid belongs to rank (cx + B * cy) mod world, a lattice whose step B is the integer
// coprime to `world` nearest to 0.618 * world (8 -> 5, 4 -> 3, 2 -> 1): every chunk column AND every chunk row is
// spread over all ranks, so a narrow expensive structure cannot land on a few of them.  Measured cost imbalance
// (slowest rank / mean, helmet and tower frames): 1.02 / 1.02 at 8 ranks against 1.07 / 1.09 for `chunk mod world`
// (which, with 60 chunk columns, gives every rank whole columns).  A rank's chunks are numbered in ascending
// global order; the tables below are the single source of that numbering for the kernels and for Python.
static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

static int partition_step(int world) {
  if (world <= 1) return 0;
  double target = 0.6180339887 * world;
  int best = 1;
  double best_d = 1e30;
  for (int b = 1; b < world; b++) {
    if (gcd_i(b, world) != 1) continue;
    double dd = b > target ? b - target : target - b;
    if (dd <= best_d) { best_d = dd; best = b; }
  }
  return best;
}

extern "C" i32 rt_chunk_owner(i32 width, i32 height, i32 world, i32 chunk) {
  i32 n = rt_chunk_count(width, height);
  if (world <= 0 || chunk < 0 || chunk >= n) return -1;
  int chunks_x = (width + RT_CHUNK_SIZE - 1) / RT_CHUNK_SIZE;
  int cx = chunk % chunks_x, cy = chunk / chunks_x;
  return (i32)((cx + (int64_t)partition_step(world) * cy) % world);
}


struct Partition {
  int width = 0, height = 0, world = 0;
  int n_chunks = 0, max_local = 0;
  std::vector<std::vector<int32_t>> lists;      // [rank] -> ascending global chunk indices
  std::vector<int32_t>              owner_slot; // [chunk] -> rank * max_local + slot
};
static std::vector<Partition *> g_partitions;
static std::mutex               g_partition_mutex;   // the host tables and every Device::parts

// The device copies of an evicted partition are not freed at once: device_chunk_list() / device_owner_table() hand out the
// pointers and release g_partition_mutex before the kernel that reads them is enqueued (under the device's own mutex), and
// another thread's 17th distinct (width, height, world) may evict in between.  They are RETIRED and freed by the eviction
// after the next one -- every frame synchronises before it returns, so whatever read them has long finished by then.
struct RetiredBuffer { int phys; void *ptr; };
static std::vector<RetiredBuffer> g_retired[2];                 // [0]: retired by the latest eviction, [1]: by the one before
static void drop_device_partitions(const Partition *q) {        // g_partition_mutex held
  for (const RetiredBuffer &rb : g_retired[1]) {
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != rb.phys) (void)hipSetDevice(rb.phys);
    (void)hipFree(rb.ptr);
    if (prev >= 0 && prev != rb.phys) (void)hipSetDevice(prev);
  }
  g_retired[1].swap(g_retired[0]);
  g_retired[0].clear();
  for (int i = 0; i < RT_MAX_DEVICES; i++) {
    Device &D = g_devs[i];
    for (size_t k = 0; k < D.parts.size(); k++) {
      if (D.parts[k].host != q) continue;
      for (int32_t *ptr : D.parts[k].d_lists) if (ptr) g_retired[0].push_back({D.phys, ptr});
      if (D.parts[k].d_owner_slot) g_retired[0].push_back({D.phys, D.parts[k].d_owner_slot});
      D.parts.erase(D.parts.begin() + (long)k);
      break;
    }
  }
}

// After rt_set_devices(): slot r >= 1 belongs to GPU (primary + r) mod count, or to the primary GPU when rehearsing.  A slot
// that was initialised under the other mapping gives back what it holds on the old GPU -- scene copies, frame buffers,
// events, partition tables -- and is initialised again by its next frame.
static void remap_device_slots() {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return;
  // A multi-device frame holds slot 0's mutex from start to end and keeps using the other slots' events, pinned buffers and
  // streams after their workers have released THEIR mutexes (gather, counter sums): tearing a slot down needs slot 0's mutex
  // first.  Lock order everywhere: slot 0 -> slot r -> g_partition_mutex.
  std::lock_guard<std::mutex> frame_lock(dev0().mutex);
  const Config c = config();
  for (int r = 1; r < RT_MAX_DEVICES; r++) {
    Device &D = g_devs[r];
    std::lock_guard<std::mutex> lock(D.mutex);
    if (!D.ready) continue;
    const int want = c.rehearse ? g_primary : (g_primary + r) % count;
    if (want == D.phys) continue;
    {
      DeviceGuard guard(D);
      (void)hipDeviceSynchronize();
      for (auto &kv : D.scene_cache) free_device_scene(kv.second);
      D.scene_cache.clear();
      D.cameras.clear();
      Workspace &W = D.ws;
      (void)hipFree(W.accum); (void)hipFree(W.image); (void)hipFree(W.linear); (void)hipFree(W.tiles); (void)hipFree(W.all_tiles);
      if (W.tiles_host) (void)hipHostFree(W.tiles_host);
      if (W.counters_host) (void)hipHostFree(W.counters_host);
      (void)hipFree(W.wave_times);
      for (hipEvent_t e : W.ev0) (void)hipEventDestroy(e);
      for (hipEvent_t e : W.ev1) (void)hipEventDestroy(e);
      for (int i = 0; i < 5; i++) if (W.ev_frame[i]) (void)hipEventDestroy(W.ev_frame[i]);
      if (D.mstream) { (void)hipStreamDestroy(D.mstream); D.mstream = nullptr; }
      W = Workspace();
      D.last_counters = nullptr;
    }
    {
      std::lock_guard<std::mutex> pl(g_partition_mutex);
      for (DevPartition &dp : D.parts) {
        for (int32_t *ptr : dp.d_lists) if (ptr) g_retired[0].push_back({D.phys, ptr});
        if (dp.d_owner_slot) g_retired[0].push_back({D.phys, dp.d_owner_slot});
      }
      D.parts.clear();
    }
    D.ready = false;
  }
}

static Partition *get_partition(int width, int height, int world) {      // g_partition_mutex held
  for (Partition *q : g_partitions)
    if (q->width == width && q->height == height && q->world == world) return q;
  if (g_partitions.size() >= 16) {              // bounded cache: forget the oldest
    Partition *old = g_partitions.front();
    drop_device_partitions(old);
    delete old;
    g_partitions.erase(g_partitions.begin());
  }
  Partition *q = new Partition();
  q->width = width; q->height = height; q->world = world;
  q->n_chunks = rt_chunk_count(width, height);
  q->lists.resize((size_t)world);
  for (int c = 0; c < q->n_chunks; c++) q->lists[(size_t)rt_chunk_owner(width, height, world, c)].push_back(c);
  for (auto &l : q->lists) if ((int)l.size() > q->max_local) q->max_local = (int)l.size();
  q->owner_slot.resize((size_t)q->n_chunks);
  for (int r = 0; r < world; r++)
    for (size_t k = 0; k < q->lists[(size_t)r].size(); k++) q->owner_slot[(size_t)q->lists[(size_t)r][k]] = r * q->max_local + (int)k;
  g_partitions.push_back(q);
  return q;
}

static bool partition_args_ok(i32 width, i32 height, i32 world) {
  return width > 0 && height > 0 && world > 0 && world <= (1 << 20) && (int64_t)width * height <= ((int64_t)1 << 28);
}

extern "C" i32 rt_local_chunk_count(i32 width, i32 height, i32 rank, i32 world) {
  if (!partition_args_ok(width, height, world) || rank < 0 || rank >= world) return 0;
  std::lock_guard<std::mutex> lock(g_partition_mutex);
  return (i32)get_partition(width, height, world)->lists[(size_t)rank].size();
}

extern "C" i32 rt_max_local_chunk_count(i32 width, i32 height, i32 world) {
  if (!partition_args_ok(width, height, world)) return 0;
  std::lock_guard<std::mutex> lock(g_partition_mutex);
  return (i32)get_partition(width, height, world)->max_local;
}

extern "C" i32 rt_local_chunk_list(i32 width, i32 height, i32 rank, i32 world, i32 *out, i32 capacity) {
  if (!partition_args_ok(width, height, world) || rank < 0 || rank >= world) return 0;
  std::lock_guard<std::mutex> lock(g_partition_mutex);
  const std::vector<int32_t> &l = get_partition(width, height, world)->lists[(size_t)rank];
  for (size_t k = 0; k < l.size() && out && (i32)k < capacity; k++) out[k] = l[k];
  return (i32)l.size();
}

static DevPartition &device_partition(Device &D, const Partition *q) {     // g_partition_mutex held
  for (DevPartition &dp : D.parts)
    if (dp.host == q) return dp;
  DevPartition dp;
  dp.host = q;
  dp.d_lists.assign((size_t)q->world, nullptr);
  D.parts.push_back(dp);
  return D.parts.back();
}

// device copy of a rank's chunk list / of the owner table on D (D's GPU is the current device)
static int device_chunk_list(Device &D, int width, int height, int rank, int world, const int32_t **d_list, int *n_local) {
  std::lock_guard<std::mutex> lock(g_partition_mutex);
  Partition *q = get_partition(width, height, world);
  DevPartition &dp = device_partition(D, q);
  const std::vector<int32_t> &l = q->lists[(size_t)rank];
  if (!dp.d_lists[(size_t)rank]) {
    int32_t *ptr = nullptr;
    HIP_TRY(hipMalloc((void **)&ptr, l.empty() ? 16 : l.size() * 4));
    if (!l.empty()) HIP_TRY(hipMemcpy(ptr, l.data(), l.size() * 4, hipMemcpyHostToDevice));
    dp.d_lists[(size_t)rank] = ptr;
  }
  *d_list = dp.d_lists[(size_t)rank];
  *n_local = (int)l.size();
  return 0;
}

static int device_owner_table(Device &D, int width, int height, int world, const int32_t **d_table, int *n_chunks) {
  std::lock_guard<std::mutex> lock(g_partition_mutex);
  Partition *q = get_partition(width, height, world);
  DevPartition &dp = device_partition(D, q);
  if (!dp.d_owner_slot) {
    HIP_TRY(hipMalloc((void **)&dp.d_owner_slot, q->owner_slot.size() * 4));
    HIP_TRY(hipMemcpy(dp.d_owner_slot, q->owner_slot.data(), q->owner_slot.size() * 4, hipMemcpyHostToDevice));
  }
  *d_table = dp.d_owner_slot;
  *n_chunks = q->n_chunks;
  return 0;
}

static int check_params(RT_Render_Params const *p) {
  if (!p) return rt_fail("render params are NULL");
  if (p->width <= 0 || p->height <= 0) return rt_fail("image size %dx%d is invalid", p->width, p->height);
  if ((int64_t)p->width * p->height > (int64_t)1 << 28) return rt_fail("image %dx%d is too large", p->width, p->height);
  if (p->samples <= 0) return rt_fail("samples must be positive (got %d)", p->samples);
  if (p->max_bounces < 0) return rt_fail("max_bounces must be >= 0 (got %d)", p->max_bounces);
  if (p->world <= 0 || p->rank < 0 || p->rank >= p->world) return rt_fail("rank %d / world %d is invalid", p->rank, p->world);
  if (p->sample_first < 0 || p->sample_count < 0 || p->sample_first + p->sample_count > p->samples)
    return rt_fail("sample range [%d, +%d) outside [0, %d)", p->sample_first, p->sample_count, p->samples);
  return 0;
}

static int fill_kparams(Device &D, RT_KParams *K, RT_Device_Scene *d, Camera const *cam, RT_Render_Params const *p,
                        void *d_accum) {
  memset(K, 0, sizeof *K);
  K->nodes = d->nodes;
  K->leaves = d->leaves;
  K->tris = d->tris;
  K->mats = d->mats;
  K->textures = d->textures;
  K->texels = d->texels;
  K->depth = d->depth;
  K->last_row_offset = d->last_row_offset;
  K->bg_texture = d->bg_texture;
  K->n_nodes = d->n_nodes;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) K->cam[i][j] = cam->view_matrix.rows[i][j];
  K->focal_length = cam->focal_length;
  {
    volatile float fw = (float)p->width, fh = (float)p->height;      // plain IEEE fp32 divisions, as the kernel used to do
    volatile float iw = 1.0f / fw, ih = 1.0f / fh, asp = fw / fh;
    K->inv_width = iw;
    K->inv_height = ih;
    K->aspect = asp;
  }
  K->width = p->width;
  K->height = p->height;
  K->samples = p->samples;
  K->max_bounces = p->max_bounces;
  K->seed = p->seed;
  K->chunks_x = (p->width + RT_CHUNK_SIZE - 1) / RT_CHUNK_SIZE;
  K->n_chunks = rt_chunk_count(p->width, p->height);
  K->rank = p->rank;
  K->world = p->world;
  {
    int n_local = 0;
    if (device_chunk_list(D, p->width, p->height, p->rank, p->world, &K->local_chunks, &n_local) != 0) return -1;
    K->n_local_chunks = n_local;
  }
  K->sample_first = p->sample_first;
  K->sample_end = p->sample_count > 0 ? p->sample_first + p->sample_count : p->samples;
  int n_samples = K->sample_end - K->sample_first;
  // (work items of the diagnostic kernel generations: samples per item = the largest of 32 / 16 / 8 that still leaves 24
  // items per wave)
  int slab = p->slab;
  if (slab <= 0) {
    slab = 8;
    for (int cand = 32; cand > 8; cand >>= 1) {
      int64_t items = (int64_t)K->n_local_chunks * 16 * ((n_samples + cand - 1) / cand);
      if (items >= (int64_t)24 * D.num_cus * 16) { slab = cand; break; }
    }
  }
  int shift = 0;
  while ((1 << shift) < slab && (1 << shift) < n_samples) shift++;
  K->slab_shift = shift;
  K->n_slabs = (n_samples + (1 << shift) - 1) >> shift;
  int64_t n_work = (int64_t)K->n_local_chunks * 16 * K->n_slabs;
  if (n_work > 0x7fffffff) return rt_fail("too many work items (%lld)", (long long)n_work);
  K->n_work = (int32_t)n_work;
  K->accum = (unsigned long long *)d_accum;
  K->counters = d->ls[0].counters;      // (render_accumulate_locked puts the launch state it was asked for)
  K->work_head = d->ls[0].work_head;
  return 0;
}

#ifdef RT_DIAG_VARIANTS
// ---- wavefront pipeline (rt_wavefront.hip; diagnostic library only) --------------------------------------------------
// Camera kernel -> (shade, trace) per bounce, joined by record queues in HBM.  The queues are sized for `cap` camera-ray
// hits per pass (grown on demand, never beyond rt_set_wavefront_capacity() records); a frame with more first hits than
// that takes several passes: the camera kernel stops taking units when its hit queue is nearly full, the bounces run, and
// the host -- which reads one control word after every pass -- launches it again; tile_next / work_head keep the position.
static int wavefront_ensure_queues(RT_Device_Scene *d, int64_t paths, int cam_waves, int max_waves) {
  const int64_t cap = g_wf_cap_records.load();
  const int64_t want = paths < cap ? paths : cap;
  // chunks: a closed chunk holds at least WF_CHUNK - 63 records; every wave leaves one open chunk behind
  const int64_t fill = WF_CHUNK - 63;
  const int64_t soft = (want + fill - 1) / fill + cam_waves + 1;
  if (d->wf_ctl && d->wf_soft0 >= soft && d->wf_waves >= max_waves) return 0;
  (void)hipFree(d->wf_hit0); (void)hipFree(d->wf_hit); (void)hipFree(d->wf_ray[0]); (void)hipFree(d->wf_ray[1]);
  (void)hipFree(d->wf_cnt); (void)hipFree(d->wf_ctl);
  d->wf_hit0 = d->wf_hit = d->wf_ray[0] = d->wf_ray[1] = d->wf_cnt = d->wf_ctl = nullptr;
  d->wf_soft0 = 0;
  const int64_t hard = soft + 2 * (int64_t)cam_waves + 8;                      // a stopped wave closes at most two more chunks
  const int64_t ray_chunks = (hard * WF_CHUNK + fill - 1) / fill + max_waves + 8;   // rays <= hits
  const int64_t hit_chunks = (ray_chunks * WF_CHUNK + fill - 1) / fill + max_waves + 8;   // hits <= rays
  HIP_TRY(hipMalloc(&d->wf_hit0, (size_t)hard * WF_HIT0_FIELDS * WF_CHUNK * 4));
  HIP_TRY(hipMalloc(&d->wf_hit, (size_t)hit_chunks * WF_HIT_FIELDS * WF_CHUNK * 4));
  HIP_TRY(hipMalloc(&d->wf_ray[0], (size_t)ray_chunks * WF_RAY_FIELDS * WF_CHUNK * 4));
  HIP_TRY(hipMalloc(&d->wf_ray[1], (size_t)ray_chunks * WF_RAY_FIELDS * WF_CHUNK * 4));
  HIP_TRY(hipMalloc(&d->wf_cnt, (size_t)(hard + hit_chunks + 2 * ray_chunks) * 4));
  HIP_TRY(hipMalloc(&d->wf_ctl, (size_t)WF_N_CTL * WF_CTL_STRIDE * 4));
  if (!d->wf_ctl_host) HIP_TRY(hipHostMalloc((void **)&d->wf_ctl_host, (size_t)WF_N_CTL * WF_CTL_STRIDE * 4, hipHostMallocDefault));
  d->wf_soft0 = soft; d->wf_hard0 = hard; d->wf_ray_chunks = ray_chunks; d->wf_hit_chunks = hit_chunks;
  d->wf_waves = max_waves;
  return 0;
}

// K: filled for the tile-stream kernel (units, tile counters, schedule feedback)
static int launch_wavefront(Device &D, RT_Device_Scene *d, RT_KParams &K, hipStream_t stream) {
  if (K.width > 65535 || K.height > 65535) return rt_fail("the wavefront pipeline packs a pixel into 16 + 16 bits: %dx%d is too large", K.width, K.height);
  int geometry = knob_int("RT_WF_GEOMETRY", 0);
  if (geometry < 0 || geometry > 2) geometry = 0;
  int geometry_cam = knob_int("RT_WF_GEOMETRY_CAM", geometry);
  if (geometry_cam < 0 || geometry_cam > 2) geometry_cam = 0;
  const int lds_limit = 160 * 1024;
  static const int wpb_of[3] = {16, 12, 10}, bpc_of[3] = {1, 2, 2};
  const int wpb_cam = wpb_of[geometry_cam], bpc_cam = bpc_of[geometry_cam], wpb_tr = wpb_of[geometry], bpc_tr = bpc_of[geometry];
  const int per_wave = (K.depth > 0 ? K.depth : 1) * 256 + 1536;                   // perm stack + accumulator tile
  const int cam_blocks = D.num_cus * bpc_cam, cam_waves = cam_blocks * wpb_cam;
  const int tr_blocks = D.num_cus * bpc_tr, tr_waves = tr_blocks * wpb_tr;
  int shade_blocks_per_cu = knob_int("RT_WF_SHADE_BLOCKS", 5);
  if (shade_blocks_per_cu < 1 || shade_blocks_per_cu > 8) shade_blocks_per_cu = 5;
  const int shade_blocks = D.num_cus * shade_blocks_per_cu, shade_waves = shade_blocks * 4;
  const int max_waves = std::max(std::max(cam_waves, tr_waves), shade_waves);
  auto lds_nodes_for = [&](int waves_per_block, int blocks_per_cu) {
    int room = (lds_limit / blocks_per_cu - waves_per_block * per_wave) / 208;
    if (room < 0) room = 0;
    int n = d->n_nodes < room ? d->n_nodes : room;
    if (!d->boxes_ordered) n = 0;
    int v = knob_int("RT_LDS_NODES", n);
    if (v >= 0 && v < n) n = v;
    return n;
  };
  const int n_lds_cam = lds_nodes_for(wpb_cam, bpc_cam), n_lds_trace = lds_nodes_for(wpb_tr, bpc_tr);
  const int smem_cam = n_lds_cam * 208 + wpb_cam * per_wave;
  const int smem_trace = n_lds_trace * 208 + wpb_tr * per_wave;

  const int64_t paths = (int64_t)K.n_tiles * 64 * (K.sample_end - K.sample_first);
  if (wavefront_ensure_queues(d, paths, cam_waves, max_waves) != 0) return -1;
  K.wf_hit0 = d->wf_hit0; K.wf_hit = d->wf_hit; K.wf_ray[0] = d->wf_ray[0]; K.wf_ray[1] = d->wf_ray[1];
  K.wf_cnt_hit0 = d->wf_cnt;
  K.wf_cnt_hit = d->wf_cnt + d->wf_hard0;
  K.wf_cnt_ray[0] = K.wf_cnt_hit + d->wf_hit_chunks;
  K.wf_cnt_ray[1] = K.wf_cnt_ray[0] + d->wf_ray_chunks;
  K.wf_ctl = d->wf_ctl;
  K.wf_soft_chunks = (int32_t)(d->wf_soft0 > 0x7fffffff ? 0x7fffffff : d->wf_soft0);
  K.park = nullptr;
  HIP_TRY(hipMemsetAsync(d->wf_ctl, 0, (size_t)WF_N_CTL * WF_CTL_STRIDE * 4, stream));

  for (int pass = 0; pass < (1 << 20); pass++) {
    if (pass > 0) HIP_TRY(hipMemsetAsync(d->wf_ctl + WF_STOPPED * WF_CTL_STRIDE, 0, 4, stream));
    K.n_lds_nodes = n_lds_cam;
    K.pyr_nodes = knob_int("RT_PYRAMID", 1) ? n_lds_cam : 0;
    K.wf_n_waves = cam_waves;
    int rc = rt_wf_launch_camera(&K, cam_blocks, geometry_cam, smem_cam, stream);
    if (rc != 0) return rt_fail("camera kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    for (int b = 0; b < K.max_bounces; b++) {
      K.wf_bounce = b;
      K.wf_n_waves = shade_waves;
      rc = rt_wf_launch_shade(&K, shade_blocks, b == 0, stream);
      if (rc != 0) return rt_fail("shade kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
      if (b + 1 >= K.max_bounces) break;
      K.wf_bounce = b + 1;
      K.wf_n_waves = tr_waves;
      K.n_lds_nodes = n_lds_trace;
      rc = rt_wf_launch_trace(&K, tr_blocks, geometry, smem_trace, stream);
      if (rc != 0) return rt_fail("trace kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
      if (b >= 7 && (b & 3) == 3) {
        // long bounce limits: most paths have ended long before; every fourth bounce look at the hit queue the next shade
        // kernel would read and stop launching when a whole bounce produced no hit
        HIP_TRY(hipMemcpyAsync(d->wf_ctl_host, d->wf_ctl, (size_t)WF_N_CTL * WF_CTL_STRIDE * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (d->wf_ctl_host[WF_HIT_ALLOC * WF_CTL_STRIDE] == 0) break;
      }
    }
    HIP_TRY(hipMemcpyAsync(d->wf_ctl_host, d->wf_ctl, (size_t)WF_N_CTL * WF_CTL_STRIDE * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (d->wf_ctl_host[WF_STOPPED * WF_CTL_STRIDE] == 0) break;
  }
  return 0;
}

#endif  // RT_DIAG_VARIANTS

// Enqueues one launch of the path tracer for p's rank / sample range.  D.mutex held, D's GPU current.
// ev_prep (optional): recorded between the per-launch preparation and the path kernel.
static int render_accumulate_locked(Device &D, RT_Device_Scene *d, Camera const *cam, RT_Render_Params const *p, void *d_accum,
                                    hipStream_t stream, hipEvent_t ev_prep = nullptr, int launch_state = 0) {
  if (ensure_device(D) != 0) return -1;
  if (check_params(p) != 0) return -1;
  if (!d || !d_accum) return rt_fail("rt_render_accumulate: NULL scene or accumulation buffer");
  if (d->dev != &D) return rt_fail("rt_render_accumulate: the scene was uploaded to another device");
  RT_KParams K;
  if (fill_kparams(D, &K, d, cam, p, d_accum) != 0) return -1;
  if (launch_state < 0 || launch_state >= RT_LAUNCH_STATES) return rt_fail("rt_render_accumulate: launch state %d out of range", launch_state);
  LaunchState &L = d->ls[launch_state];
  if (!L.counters) HIP_TRY(hipMalloc((void **)&L.counters, RT_N_COUNTERS * sizeof(unsigned long long)));
  if (!L.work_head) HIP_TRY(hipMalloc((void **)&L.work_head, 64));
  K.counters = L.counters;
  K.work_head = L.work_head;
  D.last_counters = L.counters;
  // 5 = the tile-stream kernel (the product's only generation); 1-4 exist in the diagnostic build
  int variant = knob_int("RT_KERNEL", 5);
  if (variant < 1 || variant > 5) variant = 5;
  bool wavefront = false;
#ifdef RT_DIAG_VARIANTS
  wavefront = (g_pipeline.load() == 1 || knob_is("RT_PIPELINE", "wf")) && variant == 5;
#endif

  // persistent grid: 16 waves per CU (4 per SIMD at <= 128 VGPRs), never more waves than work items
  int waves_per_cu = knob_int("RT_WAVES_PER_CU", 0);
  const bool waves_per_cu_default = waves_per_cu <= 0;
  if (waves_per_cu_default) waves_per_cu = 16;
  int wg_waves = 16;                 // waves per workgroup of the tile-stream kernel: 8 / 12 / 16, chosen below
  int n_waves = D.num_cus * waves_per_cu;
  if (n_waves > K.n_work && K.n_work > 0) n_waves = K.n_work;
  K.sched_thresh = knob_int("RT_SCHED_THRESH", 48);
  if (K.sched_thresh < 1 || K.sched_thresh > 64) K.sched_thresh = 48;
  // dynamic LDS per workgroup: per wave (perm stack: depth x 256 B, accumulator tile: 1536 B) and as many leading BVH
  // nodes (level order) as fit in the 160 KB of a CU at 208 B each
  const int lds_limit = 160 * 1024 - 64;      // (- the kernel's static LDS: the 32-byte sRGB scale table, rt_dev.hip.h rt_pow24_lds)
  int per_wave = (K.depth > 0 ? K.depth : 1) * 256 + 1536;
  int smem = 0;
  K.n_lds_nodes = 0;
  if (variant == 2) {
    smem = 4 * per_wave;
  } else if (variant != 1) {
    // Workgroup size by the size of the launch (round 5, profiles/r05_small_launch.md section 3).  A launch ends with every wave
    // running the bounce chains of its last paths on thinning lanes, and at four waves per SIMD those thin waves are ISSUE-bound:
    // with two waves per SIMD a bounce of such a chain takes half the time.  A launch with little work per wave slot is mostly
    // that tail -- config #1: 0.58 ms with 16-wave workgroups, 0.41 with 8 -- one with much work needs all four waves per SIMD
    // for its body (the driver's default frame: 2.60 / 2.89 / 3.48 ms with 16 / 12 / 8).  One workgroup per CU either way (the
    // tree fills the LDS).  Measured crossovers, in wave-fulls of paths per slot of the 16-wave grid: tower 640x360x16 (14) 0.91 /
    // 0.76 / 0.69 ms, spheres 512^2 x 16 (16) 0.83 / 0.74 / 0.75, helmet 512^2 x 16 (16) 1.23 / 1.17 / 1.41, 64 and more: 16 wins.
    int waves_per_block = 16;
    if (variant == 5) {
      const int64_t paths = (int64_t)K.n_local_chunks * 1024 * (int64_t)(K.sample_end - K.sample_first);
      int64_t per_slot = paths / ((int64_t)D.num_cus * 16 * 64);
      // (a depth-0 scene -- one leaf group, no node blocks -- traces a ray in a quarter of the instructions: its launches are as
      //  short as launches a quarter their size; quad 256^2 / 512^2 / 768^2 / 1024^2 x 64 spp: best with 8 / 12 / 12 / 16 waves,
      //  0.43 / 0.90 / 1.62 / 2.43 ms against 0.59 / 0.99 / 1.67 / 2.43 with 16, gpurun_out/r05s/wg.md)
      if (K.depth == 0) per_slot /= 4;
      waves_per_block = per_slot < 12 ? 8 : (per_slot < 40 ? 12 : 16);
      const int v = knob_int("RT_WG_WAVES", 0);
      if (v == 8 || v == 12 || v == 16) waves_per_block = v;
      if (waves_per_cu_default) waves_per_cu = waves_per_block;
      n_waves = D.num_cus * waves_per_cu;
      if (n_waves > K.n_work && K.n_work > 0) n_waves = K.n_work;
    }
    wg_waves = waves_per_block;
    int room = (lds_limit - waves_per_block * per_wave) / 208;
    if (room < 0) room = 0;
    K.n_lds_nodes = d->n_nodes < room ? d->n_nodes : room;
    if (variant == 5 && !d->boxes_ordered) K.n_lds_nodes = 0;      // (the tile-stream kernel's LDS node blocks assume min <= max)
    int v = knob_int("RT_LDS_NODES", K.n_lds_nodes);
    if (v >= 0 && v < K.n_lds_nodes) K.n_lds_nodes = v;
    smem = K.n_lds_nodes * 208 + waves_per_block * per_wave;
  }

  // ---- schedule feedback: visit expensive tiles first (costs = rays per tile of the previous launch of this view) ----
  K.order = nullptr;
  K.tile_cost = nullptr;
  K.n_tiles = K.n_local_chunks * 16;
  const uint32_t *cost_prev = nullptr;
  if (variant != 1 && !knob_is("RT_ORDER", "identity") && K.n_tiles > 0) {
    const int n_tiles = K.n_tiles;
    // the costs of a launch are reusable by a launch of the same frame shape, partition and bounce limit -- NOT only of the same
    // view: for a camera that moves between frames the previous view's costs are still a better guide than none (helmet, a rotation
    // of 0.5 / 2 / 10 degrees per frame: -0.7 / -0.8 / -0.2 % kernel time at 256 spp, -2.3 % at 1024^2 x 64 spp against the identity
    // order, tools/exp_moving.py, profiles/r05_experiments.md section 4), and an order is only ever a schedule, never a pixel
    uint64_t key = 1469598103934665603ull;
    auto mix = [&key](const void *ptr, size_t n) {
      const unsigned char *b = (const unsigned char *)ptr;
      for (size_t i = 0; i < n; i++) { key ^= b[i]; key *= 1099511628211ull; }
    };
    int32_t ids[6] = {K.width, K.height, K.rank, K.world, K.max_bounces, n_tiles};
    mix(ids, sizeof ids);
    if (L.sched_tiles != n_tiles) {
      (void)hipFree(L.cost[0]); (void)hipFree(L.cost[1]); (void)hipFree(L.order);
      L.cost[0] = L.cost[1] = L.order = nullptr;
      L.sched_tiles = 0;
      L.sched_valid = false;
      HIP_TRY(hipMalloc(&L.cost[0], (size_t)n_tiles * 4));
      HIP_TRY(hipMalloc(&L.cost[1], (size_t)n_tiles * 4));
      HIP_TRY(hipMalloc(&L.order, (size_t)n_tiles * 4));
      L.sched_tiles = n_tiles;
    }
    if (L.sched_valid && L.sched_key == key) {
      cost_prev = L.cost[L.sched_cur ^ 1];
      K.order = L.order;
    }
    K.tile_cost = L.cost[L.sched_cur];
    L.sched_cur ^= 1;                // after this launch, cost[sched_cur ^ 1] is the buffer just written
    L.sched_key = key;
    L.sched_valid = true;
  }

  if (variant == 5) {
    // unit = 2 neighbouring pixels x `slab` samples, default 64 (128 paths, pixel-major: the 64 lanes of a wave sit on one
    // pixel, then on its neighbour); it is also the granularity at which waves share a tile at the end of a launch.
    // Measured, helmet frame / rank 0 of 8: 32 samples 36.9 / 5.39 ms, 64 36.15 / 5.31, 128 36.6.
    const int n_samples = K.sample_end - K.sample_first;
    int cs = p->slab > 0 ? p->slab : 64;
    int cshift = 0;
    while ((1 << cshift) < cs && (1 << cshift) < n_samples) cshift++;
    K.chunk_shift = cshift;
    K.n_sample_blocks = (n_samples + (1 << cshift) - 1) >> cshift;
    K.n_chunks_tile = 32 * K.n_sample_blocks;          // units: 8 rows x sample blocks x 4 pixel pairs
    K.drain_thresh = knob_int("RT_DRAIN_THRESH", K.sched_thresh);
    if (K.drain_thresh < 1 || K.drain_thresh > 64) K.drain_thresh = K.sched_thresh;
    if (L.tile_next_n < K.n_tiles) {
      (void)hipFree(L.tile_next);
      L.tile_next = nullptr;
      L.tile_next_n = 0;
      // [n_tiles] chunk counters, then [ceil(n_tiles / 64)] open-tile counts of the groups
      HIP_TRY(hipMalloc(&L.tile_next, ((size_t)K.n_tiles + (size_t)((K.n_tiles + 63) / 64)) * 4));
      L.tile_next_n = K.n_tiles;
    }
    K.tile_next = L.tile_next;
    K.open_groups = L.tile_next + L.tile_next_n;
    int64_t chunks = (int64_t)K.n_tiles * K.n_chunks_tile;
    n_waves = D.num_cus * waves_per_cu;
    if ((int64_t)n_waves > chunks) n_waves = (int)chunks;
    // units per atomic: 1 unit of 128 paths (what a wave still holds when the launch runs dry is its tail); smaller
    // units (few samples) are taken in pairs.  Measured with units of 128 paths: frame 1 -> 36.25 ms, 2 -> 36.4, 4 -> 37.2.
    K.grab_max = (cshift >= 6 || (int64_t)K.n_tiles < (int64_t)2 * n_waves) ? 1 : 2;
    {
      int v = knob_int("RT_GRAB", 0);
      if (v == 1 || v == 2 || v == 4) K.grab_max = v;
    }
    K.pyr_nodes = knob_int("RT_PYRAMID", 1) ? K.n_lds_nodes : 0;
    // Leaf blocks with the short reciprocal (rcp_exact, rt_dev.hip.h): equal to the IEEE division while every triangle
    // determinant |e1 . (d x e2)| <= 6 D E^2 stays below 2^102.  E = largest edge component of the scene; D = largest
    // component of a ray direction: <= 3 max|view matrix entry| for camera rays (the direction is normalised before the
    // matrix is applied), < 2 for the normalised directions that shading emits.  E <= 2^38 and matrix entries <= 2^16
    // give 6 D E^2 < 2^97.  Anything else -- or a NaN -- renders with the kernel that divides.
    float cam_max = 0.0f;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        float m = fabsf(K.cam[i][j]);
        if (!(m <= cam_max)) cam_max = m;
      }
    K.short_div = (d->max_edge <= 0x1p38f && cam_max <= 0x1p16f) ? 1 : 0;
    if (knob_int("RT_SHORT_DIV", 1) == 0) K.short_div = 0;
    // hits parked until a dense shade block can be made of them: RT_PARK_RECORD_DWORDS = 18 fields x 128 records per wave
    K.park = nullptr;
    if (!wavefront && knob_int("RT_PARK", 1) != 0 && K.max_bounces < (1 << 26)) {      // (a parked record keeps the bounce count in 26 bits)
      const int grid_waves = (n_waves + wg_waves - 1) / wg_waves * wg_waves;         // whole workgroups are launched
      if (L.park_waves < grid_waves) {
        (void)hipFree(L.park);
        L.park = nullptr;
        L.park_waves = 0;
        const size_t slice_bytes = (size_t)RT_PARK_RECORD_DWORDS * 4;      // a wave's slice: 18 fields x 128 records (rt_device.h)
        HIP_TRY(hipMalloc(&L.park, (size_t)grid_waves * slice_bytes));
        L.park_waves = grid_waves;
      }
      K.park = L.park;
    }
  }

  // ---- ONE preparation launch: counters, work head, tile / unit counters, this launch's cost buffer, tile order ----
  {
    int rc2 = rt_launch_prepare(K.n_tiles, variant == 5 ? K.tile_next : nullptr, variant == 5 ? K.open_groups : nullptr, L.counters,
                                L.work_head, K.tile_cost, cost_prev, cost_prev ? L.order : nullptr, stream);
    if (rc2 != 0) return rt_fail("prepare kernel launch failed: %s", hipGetErrorString((hipError_t)rc2));
  }
  if (ev_prep) HIP_TRY(hipEventRecord(ev_prep, stream));
  if (K.n_work == 0) return 0;

  K.wave_times = nullptr;
  if (variant == 4 || (variant == 5 && knob_set("RT_WAVE_TIMES"))) {      // wave timeline (tools/exp_waves.py)
    if (!D.ws.wave_times) HIP_TRY(hipMalloc(&D.ws.wave_times, (size_t)65536 * 3 * 8));
    HIP_TRY(hipMemsetAsync(D.ws.wave_times, 0, (size_t)65536 * 3 * 8, stream));
    K.wave_times = D.ws.wave_times;
    if (n_waves > 65536) n_waves = 65536;          // the diagnostic buffer holds that many waves
    D.ws.wave_times_n = n_waves;
  }

  size_t slot = D.ws.n_timed % RT_MAX_TIMED;
  if (slot >= D.ws.ev0.size()) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    D.ws.ev0.push_back(a);
    D.ws.ev1.push_back(b);
  }
  HIP_TRY(hipEventRecord(D.ws.ev0[slot], stream));
#ifdef RT_DIAG_VARIANTS
  if (wavefront) {
    if (launch_wavefront(D, d, K, stream) != 0) return -1;
  } else
#endif
  {
    int rc = rt_launch_path_kernel(&K, n_waves, variant, smem, wg_waves, stream);
    if (rc != 0) return rt_fail("path kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  }
  HIP_TRY(hipEventRecord(D.ws.ev1[slot], stream));
  D.ws.n_timed += 1;
  return 0;
}

extern "C" int rt_set_camera(RT_Device_Scene *dscene, Camera const *camera) {
  if (!dscene || !camera) return rt_fail("rt_set_camera: NULL argument");
  Device &D = *dscene->dev;
  std::lock_guard<std::mutex> lock(D.mutex);
  D.cameras[dscene] = *camera;
  return 0;
}

extern "C" int rt_render_accumulate(RT_Device_Scene *dscene, RT_Render_Params const *params, void *d_accum,
                                    void *stream) {
  if (!dscene) return rt_fail("rt_render_accumulate: NULL scene or accumulation buffer");
  Device &D = *dscene->dev;
  std::lock_guard<std::mutex> lock(D.mutex);
  DeviceGuard guard(D);
  auto it = D.cameras.find(dscene);
  if (it == D.cameras.end()) return rt_fail("rt_render_accumulate: no camera set for this scene (rt_set_camera)");
  {
    std::lock_guard<std::mutex> lk(g_multi_mutex);           // rt_get_counters() now means THIS launch, not an older multi-device frame
    g_multi_counters_valid = false;
  }
  return render_accumulate_locked(D, dscene, &it->second, params, d_accum, (hipStream_t)stream);
}

static int resolve_on(Device &D, RT_Render_Params const *p, void const *d_accum, void *d_tiles, void *d_image, void *d_linear,
                      hipStream_t stream) {
  if (check_params(p) != 0) return -1;
  if (!d_accum) return rt_fail("rt_resolve: NULL accumulation buffer");
  int chunks_x = (p->width + RT_CHUNK_SIZE - 1) / RT_CHUNK_SIZE;
  const int32_t *d_list = nullptr;
  int n_local = 0;
  if (device_chunk_list(D, p->width, p->height, p->rank, p->world, &d_list, &n_local) != 0) return -1;
  int rc = rt_launch_resolve(p->width, p->height, p->samples, chunks_x, d_list, n_local,
                             (const unsigned long long *)d_accum, (uint8_t *)d_tiles, (uint8_t *)d_image,
                             (float *)d_linear, stream);
  if (rc != 0) return rt_fail("resolve kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

extern "C" int rt_resolve(RT_Render_Params const *p, void const *d_accum, void *d_tiles, void *d_image,
                          void *d_linear, void *stream) {
  Device &D = dev0();
  {
    std::lock_guard<std::mutex> lock(D.mutex);
    if (ensure_device(D) != 0) return -1;
  }
  return resolve_on(D, p, d_accum, d_tiles, d_image, d_linear, (hipStream_t)stream);
}

static int untile_on(Device &D, i32 width, i32 height, i32 world, void const *d_all_tiles, void *d_image, hipStream_t stream) {
  if (width <= 0 || height <= 0 || world <= 0 || !d_all_tiles || !d_image) return rt_fail("rt_untile: bad arguments");
  int chunks_x = (width + RT_CHUNK_SIZE - 1) / RT_CHUNK_SIZE;
  const int32_t *d_table = nullptr;
  int n_chunks = 0;
  if (!partition_args_ok(width, height, world)) return rt_fail("rt_untile: bad arguments");
  if (device_owner_table(D, width, height, world, &d_table, &n_chunks) != 0) return -1;
  int rc = rt_launch_untile(width, height, chunks_x, n_chunks, d_table, (const uint8_t *)d_all_tiles,
                            (uint8_t *)d_image, stream);
  if (rc != 0) return rt_fail("untile kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

extern "C" int rt_untile(i32 width, i32 height, i32 world, void const *d_all_tiles, void *d_image, void *stream) {
  Device &D = dev0();
  {
    std::lock_guard<std::mutex> lock(D.mutex);
    if (ensure_device(D) != 0) return -1;
  }
  return untile_on(D, width, height, world, d_all_tiles, d_image, (hipStream_t)stream);
}

static int ensure_ws_buffers(Workspace &W, int width, int height, size_t tiles_bytes, size_t all_tiles_bytes, bool want_linear = true) {
  size_t pixels = (size_t)width * height;
  if (W.accum_elems < pixels * 3) {
    (void)hipFree(W.accum);
    W.accum = nullptr;
    W.accum_elems = 0;
    HIP_TRY(hipMalloc(&W.accum, pixels * 3 * sizeof(unsigned long long)));
    W.accum_elems = pixels * 3;
  }
  if (W.image_pixels < pixels) {
    (void)hipFree(W.image);
    (void)hipFree(W.linear);
    W.image = nullptr;
    W.linear = nullptr;
    W.image_pixels = 0;
    HIP_TRY(hipMalloc(&W.image, pixels * 3));
    if (want_linear) HIP_TRY(hipMalloc(&W.linear, pixels * 3 * sizeof(float)));      // (a frame lane has no fp32 output)
    W.image_pixels = pixels;
  }
  if (W.tiles_bytes < tiles_bytes) {
    (void)hipFree(W.tiles);
    W.tiles = nullptr;
    W.tiles_bytes = 0;
    HIP_TRY(hipMalloc(&W.tiles, tiles_bytes));
    W.tiles_bytes = tiles_bytes;
  }
  if (W.all_tiles_bytes < all_tiles_bytes) {
    (void)hipFree(W.all_tiles);
    W.all_tiles = nullptr;
    W.all_tiles_bytes = 0;
    HIP_TRY(hipMalloc(&W.all_tiles, all_tiles_bytes));
    W.all_tiles_bytes = all_tiles_bytes;
  }
  for (int i = 0; i < 5; i++)
    if (!W.ev_frame[i]) HIP_TRY(hipEventCreate(&W.ev_frame[i]));
  return 0;
}

static int ensure_frame_buffers(Device &D, int width, int height, size_t tiles_bytes, size_t all_tiles_bytes) {
  return ensure_ws_buffers(D.ws, width, height, tiles_bytes, all_tiles_bytes);
}

static int copy_image_out(Image const *image, const uint8_t *d_image, int width, int height, hipStream_t stream) {
  size_t pixels = (size_t)width * height;
  if (!image->pixels.data) return 0;
  if (image->components == 3 && image->stride == image->width) {
    HIP_TRY(hipMemcpyAsync(image->pixels.data, d_image, pixels * 3, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
  } else {
    std::vector<uint8_t> tmp(pixels * 3);
    HIP_TRY(hipMemcpy(tmp.data(), d_image, pixels * 3, hipMemcpyDeviceToHost));
    for (isize y = 0; y < image->height; y++)
      for (isize x = 0; x < image->width; x++)
        for (int c = 0; c < 3; c++)
          image->pixels.data[image->components * (x + y * image->stride) + c] = tmp[((size_t)y * width + x) * 3 + c];
  }
  return 0;
}

static float event_ms(hipEvent_t a, hipEvent_t b) {
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.0f;
  return ms;
}

// ---- a frame spread over the N GPUs of this node, behind the reference's own entry points ------------------------------
// driver.c:793-803 starts `-T n` threads on render_thread_proc; the one that claims the frame drives N devices (N =
// rt_device_count()): every device holds its own copy of the scene, renders the chunks rt_chunk_owner() gives its rank and
// resolves them to compact u8 tiles; the tiles travel to device 0 with ONE peer copy per device over xGMI
// (hipMemcpyPeerAsync; 0.78 MB each at 1080p), device 0 untiles into the row-major image and copies it to the caller's
// pixels.  Devices are driven by one internal host thread each (uploads and launches proceed in parallel; the calling
// application's other threads return at once, as in the one-GPU case).  Per-path seeds depend on (pixel, sample) only, so
// the image does not depend on N (tests/test_gpu_multi_device.py: byte-equal to the one-device frame).
// How the N devices are driven (round 4):
//  * one PERSISTENT internal host thread per device slot >= 1, started the first time the slot is used and parked on a
//    condition variable between frames (round 3 created and joined N - 1 threads per frame);
//  * a device thread only ENQUEUES -- clear, preparation, path kernel, resolve to compact tiles, the tile copy towards slot 0,
//    an asynchronous copy of its ray counters into pinned host memory, an event -- and returns; nothing on it waits for the GPU;
//  * slot 0's stream waits for the other devices' events (hipStreamWaitEvent), untiles, copies the image out: ONE host
//    synchronisation per frame, at the end;
//  * a device without peer access to slot 0 (hipDeviceCanAccessPeer refused) stages its tiles through pinned host memory;
//  * the full content check of the host scene (rt_scene_touch) runs on the calling thread while all of that is in flight;
//  * rt_get_frame_timing() reports the slowest device's prep / path / resolve / copy times, the gather + untile + copy-out
//    time after it, and which device it was.
struct MultiFrame {
  Scene const      *scene = nullptr;
  RT_Render_Params  base;
  int               world = 0, w = 0, h = 0;
  size_t            tiles_bytes = 0;
  int               rcs[RT_MAX_DEVICES];
  bool              staged[RT_MAX_DEVICES];
  float             stamp_ms[RT_MAX_DEVICES], upload_ms[RT_MAX_DEVICES], enqueue_ms[RT_MAX_DEVICES];
  uint64_t          full_fp[RT_MAX_DEVICES];       // the full fingerprint of the copy each slot rendered from
  char              err[RT_MAX_DEVICES][256];
  std::mutex              m;
  std::condition_variable cv;
  int                     pending = 0;
};

struct DevWorker {
  std::mutex              m;
  std::condition_variable cv;
  MultiFrame             *job = nullptr;
  bool                    started = false;
};
static DevWorker *g_workers = new DevWorker[RT_MAX_DEVICES];     // never destroyed: the parked threads outlive static destructors

#ifdef RT_DIAG_VARIANTS
// fault injection for tests/test_gpu_multi_device.py (diagnostic library only): bit 0 = pretend no device has peer access
// to slot 0 (staged tile copies), bits 8.. = 1 + the slot whose frame fails
static std::atomic<int> g_multi_fault{0};
extern "C" void rt_diag_multi_fault(i32 no_peer, i32 failing_slot) {
  g_multi_fault.store((no_peer ? 1 : 0) | ((failing_slot >= 0 ? failing_slot + 1 : 0) << 8));
}
static bool fault_no_peer() { return (g_multi_fault.load() & 1) != 0; }
static bool fault_fails(int slot) { return (g_multi_fault.load() >> 8) == slot + 1; }
#else
static inline bool fault_no_peer() { return false; }
static inline bool fault_fails(int) { return false; }
#endif

static void device_fail(MultiFrame &J, int r, const char *what) {
  Device &D = g_devs[r];
  snprintf(J.err[r], sizeof J.err[r], "device %d (slot %d of %d): %s", D.phys, r, J.world, what);
}

// Everything device slot `r` contributes to the frame, enqueued on its null stream; returns without waiting for the GPU.
static void enqueue_device_frame(MultiFrame &J, int r) {
  Device &D = g_devs[r];
  Device &D0 = dev0();
  std::unique_lock<std::mutex> lock(D.mutex, std::defer_lock);
  if (r != 0) lock.lock();                                   // (slot 0's mutex is held by the frame's owner)
  J.rcs[r] = -1;
  J.staged[r] = false;
  J.err[r][0] = 0;
  J.full_fp[r] = 0;
  D.slot = r;
  if (ensure_device(D) != 0) { device_fail(J, r, rt_last_error()); return; }       // assigns D.phys and makes it this thread's device
  if (fault_fails(r)) { device_fail(J, r, "injected failure (rt_diag_multi_fault)"); return; }
  RT_Device_Scene *d = cached_scene_locked(D, J.scene, &J.stamp_ms[r], &J.upload_ms[r]);
  if (!d) { device_fail(J, r, rt_last_error()); return; }
  J.full_fp[r] = d->full_fp;                                 // what THIS slot's copy was made from
  const double t_enq = now_ms();
  Workspace &W = D.ws;
  if (ensure_frame_buffers(D, J.w, J.h, J.tiles_bytes, r == 0 ? J.tiles_bytes * (size_t)J.world : 0) != 0) { device_fail(J, r, rt_last_error()); return; }
  if (!W.counters_host && hipHostMalloc((void **)&W.counters_host, RT_N_COUNTERS * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
    device_fail(J, r, "pinned counter buffer"); return;
  }
  const bool same_gpu = D.phys == D0.phys;
  const bool direct = same_gpu || (D.peer_ok && !fault_no_peer());
  if (!direct && W.tiles_host_bytes < J.tiles_bytes) {
    if (W.tiles_host) (void)hipHostFree(W.tiles_host);
    W.tiles_host = nullptr; W.tiles_host_bytes = 0;
    if (hipHostMalloc((void **)&W.tiles_host, J.tiles_bytes, hipHostMallocDefault) != hipSuccess) { device_fail(J, r, "pinned tile buffer"); return; }
    W.tiles_host_bytes = J.tiles_bytes;
  }
  RT_Render_Params p = J.base;
  p.rank = r;
  p.world = J.world;
  // every slot works on a stream of its own: on N GPUs that changes nothing; N slots rehearsed on ONE GPU overlap -- the
  // next slot's workgroups take the CUs the previous slot's last paths leave idle -- instead of queueing N launch tails
  hipStream_t stream = D.mstream;
  hipError_t e = hipEventRecord(W.ev_frame[0], stream);
  if (e == hipSuccess) e = hipMemsetAsync(W.accum, 0, (size_t)J.w * J.h * 3 * sizeof(unsigned long long), stream);
  if (e != hipSuccess) { device_fail(J, r, hipGetErrorString(e)); return; }
  if (render_accumulate_locked(D, d, &J.scene->camera, &p, W.accum, stream, W.ev_frame[1]) != 0) { device_fail(J, r, rt_last_error()); return; }
  (void)hipEventRecord(W.ev_frame[2], stream);
  if (resolve_on(D, &p, W.accum, W.tiles, nullptr, nullptr, stream) != 0) { device_fail(J, r, rt_last_error()); return; }
  (void)hipEventRecord(W.ev_frame[3], stream);
  uint8_t *dst = D0.ws.all_tiles + (size_t)r * J.tiles_bytes;
  if (same_gpu) e = hipMemcpyAsync(dst, W.tiles, J.tiles_bytes, hipMemcpyDeviceToDevice, stream);
  else if (direct) e = hipMemcpyPeerAsync(dst, D0.phys, W.tiles, D.phys, J.tiles_bytes, stream);      // one xGMI link per device
  else { e = hipMemcpyAsync(W.tiles_host, W.tiles, J.tiles_bytes, hipMemcpyDeviceToHost, stream); J.staged[r] = true; }
  if (e == hipSuccess) e = hipMemcpyAsync(W.counters_host, d->ls[0].counters, RT_N_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipEventRecord(W.ev_frame[4], stream);
  if (e != hipSuccess) { device_fail(J, r, hipGetErrorString(e)); return; }
  J.enqueue_ms[r] = (float)(now_ms() - t_enq);
  J.rcs[r] = 0;
}

static void worker_loop(int r) {
  DevWorker &Wk = g_workers[r];
  for (;;) {
    MultiFrame *J;
    {
      std::unique_lock<std::mutex> lk(Wk.m);
      Wk.cv.wait(lk, [&] { return Wk.job != nullptr; });
      J = Wk.job;
      Wk.job = nullptr;
    }
    enqueue_device_frame(*J, r);
    {
      std::lock_guard<std::mutex> lk(J->m);
      J->pending -= 1;
    }
    J->cv.notify_all();
  }
}

static void drop_scene_everywhere_but0(Scene const *scene, int world) {
  for (int r = 1; r < world; r++) {
    Device &D = g_devs[r];
    std::lock_guard<std::mutex> lock(D.mutex);
    auto it = D.scene_cache.find(scene);
    if (it == D.scene_cache.end()) continue;
    DeviceGuard guard(D);
    (void)hipDeviceSynchronize();
    free_device_scene(it->second);
    D.scene_cache.erase(it);
  }
}

static int render_frame_multi(Scene const *scene, Image const *image, RT_Render_Params base, int world) {
  Device &D0 = dev0();                               // D0.mutex held by the caller
  const double t0 = now_ms();
  const int w = base.width, h = base.height;
  const size_t max_local = (size_t)rt_max_local_chunk_count(w, h, world);
  const size_t tiles_bytes = max_local * 1024 * 3;
  if (ensure_frame_buffers(D0, w, h, tiles_bytes, tiles_bytes * (size_t)world) != 0) return -1;
  const bool verify = !scene_is_static(scene);
  FrameTiming T;
  T.n_devices = world;
  static MultiFrame J;                               // (one multi-device frame at a time: D0.mutex)
  for (int attempt = 0;; attempt++) {
    J.scene = scene; J.base = base; J.world = world; J.w = w; J.h = h; J.tiles_bytes = tiles_bytes;
    for (int r = 0; r < world; r++) { J.stamp_ms[r] = J.upload_ms[r] = J.enqueue_ms[r] = 0.0f; J.rcs[r] = -1; }
    {
      std::lock_guard<std::mutex> lk(J.m);
      J.pending = world - 1;
    }
    for (int r = 1; r < world; r++) {
      DevWorker &Wk = g_workers[r];
      {
        std::lock_guard<std::mutex> lk(Wk.m);
        if (!Wk.started) { std::thread(worker_loop, r).detach(); Wk.started = true; }
        Wk.job = &J;
      }
      Wk.cv.notify_one();
    }
    enqueue_device_frame(J, 0);
    // the full content check of the host scene while every device renders (see rt_scene_touch)
    bool stale = false;
    uint64_t fp_now = 0;
    if (verify && attempt == 0) {
      const double t_v = now_ms();
      fp_now = scene_fingerprint(scene);
      T.verify_ms = (float)(now_ms() - t_v);
    }
    {
      std::unique_lock<std::mutex> lk(J.m);
      J.cv.wait(lk, [&] { return J.pending == 0; });
    }
    int failed = -1;
    for (int r = 0; r < world; r++)
      if (J.rcs[r] != 0 && failed < 0) failed = r;
    // every slot against the fingerprint of ITS OWN copy: slot r >= 1 may hold a copy made before an in-place edit that
    // slot 0 has long re-uploaded (one fingerprint per Scene* let such a copy through)
    if (verify && attempt == 0)
      for (int r = 0; r < world; r++)
        if (J.rcs[r] == 0 && J.full_fp[r] != fp_now) stale = true;
    if (failed >= 0 || stale) {
      // nothing may still be writing into slot 0's tile buffer, or reading a scene copy, when we leave or start over
      for (int r = 0; r < world; r++)
        if (J.rcs[r] == 0) (void)hipEventSynchronize(g_devs[r].ws.ev_frame[4]);
      if (failed >= 0) return rt_fail("multi-device frame failed: %s", J.err[failed][0] ? J.err[failed] : "unknown error");
      {                                               // drop the copies that do not match the host scene as it is now
        auto it = D0.scene_cache.find(scene);
        if (it != D0.scene_cache.end() && it->second->full_fp != fp_now) { free_device_scene(it->second); D0.scene_cache.erase(it); }
        drop_stale_copies(scene, fp_now, 1);
      }
      continue;                                       // the host scene changed under the cached copies: upload and render again
    }
    break;
  }
  DeviceGuard guard(D0);
  hipStream_t s0 = D0.mstream;
  const double t_g = now_ms();
  for (int r = 1; r < world; r++) {
    Workspace &W = g_devs[r].ws;
    if (J.staged[r]) {                                // no peer access: pinned host memory -> slot 0
      HIP_TRY(hipEventSynchronize(W.ev_frame[4]));
      HIP_TRY(hipMemcpyAsync(D0.ws.all_tiles + (size_t)r * tiles_bytes, W.tiles_host, tiles_bytes, hipMemcpyHostToDevice, s0));
    } else {
      HIP_TRY(hipStreamWaitEvent(s0, W.ev_frame[4], 0));
    }
  }
  if (untile_on(D0, w, h, world, D0.ws.all_tiles, D0.ws.image, s0) != 0) return -1;
  if (copy_image_out(image, D0.ws.image, w, h, s0) != 0) return -1;
  HIP_TRY(hipStreamSynchronize(s0));
  // the slowest device's split, the ray counters of all
  float worst = -1.0f;
  RT_Counters sum;
  memset(&sum, 0, sizeof sum);
  for (int r = 0; r < world; r++) {
    Workspace &W = g_devs[r].ws;
    const float span = event_ms(W.ev_frame[0], W.ev_frame[4]);
    if (span > worst) {
      worst = span;
      T.slowest_device = r;
      T.gpu_prep_ms = event_ms(W.ev_frame[0], W.ev_frame[1]);
      T.gpu_path_ms = event_ms(W.ev_frame[1], W.ev_frame[2]);
      T.gpu_resolve_ms = event_ms(W.ev_frame[2], W.ev_frame[3]);
      T.gpu_copy_ms = event_ms(W.ev_frame[3], W.ev_frame[4]);         // its tiles to slot 0 (+ counters)
    }
    T.stamp_ms = J.stamp_ms[r] > T.stamp_ms ? J.stamp_ms[r] : T.stamp_ms;
    T.upload_ms = J.upload_ms[r] > T.upload_ms ? J.upload_ms[r] : T.upload_ms;
    T.enqueue_ms = J.enqueue_ms[r] > T.enqueue_ms ? J.enqueue_ms[r] : T.enqueue_ms;
    const unsigned long long *c = W.counters_host;
    sum.paths += c[0]; sum.rays += c[1]; sum.node_visits += c[2]; sum.leaf_visits += c[3]; sum.shades += c[4]; sum.backgrounds += c[5];
    sum.textured += c[6];
  }
  T.gather_ms = (float)(now_ms() - t_g);
  {
    std::lock_guard<std::mutex> lk(g_multi_mutex);
    g_multi_counters = sum;
    g_multi_counters_valid = true;
  }
  T.total_ms = (float)(now_ms() - t0);
  D0.timing = T;
  return 0;
}

static int render_frame_locked(Scene const *scene, Image const *image, isize samples, isize max_bounces,
                               f32 *linear, u64 *accum, Camera const *camera = nullptr, u32 const *seed = nullptr) {
  Device &D = dev0();
  const double t_start = now_ms();
  if (ensure_device(D) != 0) return -1;
  if (!scene || !image) return rt_fail("render: NULL scene or image");
  if (image->pixels.data && image->components < 3) return rt_fail("render: image needs >= 3 components");
  if (image->pixels.data && image->stride < image->width) return rt_fail("render: image stride < width");
  RT_Render_Params p;
  memset(&p, 0, sizeof p);
  p.width = (i32)image->width;
  p.height = (i32)image->height;
  p.samples = (i32)samples;
  p.max_bounces = (i32)max_bounces;
  p.seed = seed ? *seed : g_seed.load();      // (camera / seed given: a frame of rt_frame_begin rendered again, as it was begun)
  p.rank = 0;
  p.world = 1;
  if (check_params(&p) != 0) return -1;
  {
    std::lock_guard<std::mutex> lk(g_multi_mutex);
    g_multi_counters_valid = false;
  }
  const int world = rt_device_count();
  if (world > 1 && !linear && !accum && rt_chunk_count(p.width, p.height) >= world)
    return render_frame_multi(scene, image, p, world);

  FrameTiming T;
  if (ensure_frame_buffers(D, p.width, p.height, 0, 0) != 0) return -1;
  Workspace &W = D.ws;
  size_t pixels = (size_t)p.width * p.height;
  hipStream_t stream = nullptr;
  const bool verify = !scene_is_static(scene);
  for (int attempt = 0;; attempt++) {
    float stamp_ms = 0.0f, upload_ms = 0.0f;
    RT_Device_Scene *d = cached_scene_locked(D, scene, &stamp_ms, &upload_ms);
    if (!d) return -1;
    T.stamp_ms += stamp_ms;
    T.upload_ms += upload_ms;
    const double t_enq = now_ms();
    HIP_TRY(hipEventRecord(W.ev_frame[0], stream));
    HIP_TRY(hipMemsetAsync(W.accum, 0, pixels * 3 * sizeof(unsigned long long), stream));
    if (render_accumulate_locked(D, d, camera ? camera : &scene->camera, &p, W.accum, stream, W.ev_frame[1]) != 0) return -1;
    HIP_TRY(hipEventRecord(W.ev_frame[2], stream));
    if (resolve_on(D, &p, W.accum, nullptr, W.image, linear ? W.linear : nullptr, stream) != 0) return -1;
    HIP_TRY(hipEventRecord(W.ev_frame[3], stream));
    T.enqueue_ms = (float)(now_ms() - t_enq);
    // the full content check of the host scene, on this thread, while the GPU renders (see rt_scene_touch): a frame of an
    // unchanged scene waits for max(kernel, check) instead of kernel + check; a changed scene is uploaded and rendered again
    if (!verify || attempt > 0 || upload_ms > 0.0f) break;
    const double t_v = now_ms();
    const bool same = scene_fingerprint(scene) == d->full_fp;
    T.verify_ms = (float)(now_ms() - t_v);
    if (same) break;
    HIP_TRY(hipStreamSynchronize(stream));
    free_device_scene(d);
    D.scene_cache.erase(scene);
  }

  if (copy_image_out(image, W.image, p.width, p.height, stream) != 0) return -1;
  HIP_TRY(hipEventRecord(W.ev_frame[4], stream));
  if (linear) HIP_TRY(hipMemcpy(linear, W.linear, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost));
  if (accum) HIP_TRY(hipMemcpy(accum, W.accum, pixels * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipStreamSynchronize(stream));
  HIP_TRY(hipGetLastError());
  T.gpu_prep_ms = event_ms(W.ev_frame[0], W.ev_frame[1]);
  T.gpu_path_ms = event_ms(W.ev_frame[1], W.ev_frame[2]);
  T.gpu_resolve_ms = event_ms(W.ev_frame[2], W.ev_frame[3]);
  T.gpu_copy_ms = event_ms(W.ev_frame[3], W.ev_frame[4]);
  T.total_ms = (float)(now_ms() - t_start);
  D.timing = T;
  return 0;
}

extern "C" int rt_render_frame(Scene const *scene, Image const *image, isize samples, isize max_bounces, f32 *linear,
                               u64 *accum) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  return render_frame_locked(scene, image, samples, max_bounces, linear, accum);
}

// A lane's stream must not share a HARDWARE queue with the other lane's: the runtime multiplexes its streams onto a few HSA queues
// (GPU_MAX_HW_QUEUES, 4 by default), and two streams on one queue run their kernels one after the other -- measured: with ONE more
// stream in the process (a torch side stream) two plain non-blocking streams landed on one queue and the overlap was gone (2.67
// instead of 2.23 ms per default frame, gpurun_out/r05fl).  The runtime pools its queues per stream priority, so the lanes take
// different priorities: never the same queue, whatever else the process creates.  (A stream with an all-ones CU mask owns its queue
// too and measures the same, but it is a blocking stream: it would wait for every null-stream operation of the host.)
static hipError_t create_lane_stream(hipStream_t *s, int lane) {
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, lane == 0 ? 0 : hi);
}

// ---- frames in flight (rt_hip.h) ---------------------------------------------------------------------------------------------
extern "C" int rt_frame_begin(Scene const *scene, Image const *image, isize samples, isize max_bounces) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  const double t_start = now_ms();
  if (ensure_device(D) != 0) return -1;
  if (!scene || !image) return rt_fail("rt_frame_begin: NULL scene or image");
  if (image->pixels.data && image->components < 3) return rt_fail("rt_frame_begin: image needs >= 3 components");
  if (image->pixels.data && image->stride < image->width) return rt_fail("rt_frame_begin: image stride < width");
  RT_Render_Params p;
  memset(&p, 0, sizeof p);
  p.width = (i32)image->width;
  p.height = (i32)image->height;
  p.samples = (i32)samples;
  p.max_bounces = (i32)max_bounces;
  p.seed = g_seed.load();
  p.rank = 0;
  p.world = 1;
  if (check_params(&p) != 0) return -1;
  int ticket = -1;
  for (int k = 0; k < RT_FRAME_LANES; k++)
    if (!D.lanes[k].busy) { ticket = k; break; }
  if (ticket < 0) return rt_fail("rt_frame_begin: %d frames are in flight already (rt_frame_end one of them first)", RT_FRAME_LANES);
  FrameLane &F = D.lanes[ticket];
  F.scene = scene; F.image = *image; F.p = p; F.camera = scene->camera; F.d = nullptr; F.fp = 0; F.rc = 0; F.finished = false; F.timing = FrameTiming();
  F.t_begin = t_start;
  if (rt_device_count() > 1) {
    // a frame over N devices has its own pipeline (render_frame_multi): rendered here and now, rt_frame_end() reports how it went
    F.rc = render_frame_locked(scene, image, samples, max_bounces, nullptr, nullptr);
    F.finished = true;
    F.busy = true;
    return ticket;
  }
  {
    std::lock_guard<std::mutex> lk(g_multi_mutex);
    g_multi_counters_valid = false;
  }
  if (!F.stream) HIP_TRY(create_lane_stream(&F.stream, ticket));
  if (ensure_ws_buffers(F.ws, p.width, p.height, 0, 0, false) != 0) return -1;
  Workspace &W = F.ws;
  const size_t pixels = (size_t)p.width * p.height;
  RT_Device_Scene *d = cached_scene_locked(D, scene, &F.timing.stamp_ms, &F.timing.upload_ms);
  if (!d) return -1;
  const double t_enq = now_ms();
  HIP_TRY(hipEventRecord(W.ev_frame[0], F.stream));
  HIP_TRY(hipMemsetAsync(W.accum, 0, pixels * 3 * sizeof(unsigned long long), F.stream));
  if (render_accumulate_locked(D, d, &F.camera, &p, W.accum, F.stream, W.ev_frame[1], 1 + ticket) != 0) return -1;
  HIP_TRY(hipEventRecord(W.ev_frame[2], F.stream));
  if (resolve_on(D, &p, W.accum, nullptr, W.image, nullptr, F.stream) != 0) return -1;
  HIP_TRY(hipEventRecord(W.ev_frame[3], F.stream));
  F.timing.enqueue_ms = (float)(now_ms() - t_enq);
  F.d = d;
  F.fp = d->full_fp;
  F.verify = !scene_is_static(scene) && F.timing.upload_ms == 0.0f;      // (a copy made for this frame IS the host scene)
  F.busy = true;
  return ticket;
}

extern "C" int rt_frame_end(int ticket) {
  Device &D = dev0();
  std::unique_lock<std::mutex> lock(D.mutex);
  if (ticket < 0 || ticket >= RT_FRAME_LANES || !D.lanes[ticket].busy || D.lanes[ticket].ending)
    return rt_fail("rt_frame_end: no frame in flight with ticket %d", ticket);
  FrameLane &F = D.lanes[ticket];
  if (F.finished) { F.busy = false; return F.rc; }
  if (ensure_device(D) != 0) { F.busy = false; return -1; }
  Workspace &W = F.ws;
  // The wait happens WITHOUT the device's mutex: another host thread can begin (or end) the other lane's frame, or render a
  // blocking one, meanwhile.  The lane stays busy -- nobody else touches it -- and `ending` refuses a second end of this ticket.
  F.ending = true;
  hipStream_t stream = F.stream;
  const bool verify = F.verify;
  Scene const *scene = F.scene;
  lock.unlock();
  // the full content check of the blocking path (render_frame_locked), on this thread, while the GPU renders: the frame came from
  // a copy with fingerprint F.fp; a host scene that no longer has it is rendered again, like there
  uint64_t now = 0;
  float verify_ms = 0.0f;
  if (verify) {
    const double t_v = now_ms();
    now = scene_fingerprint(scene);
    verify_ms = (float)(now_ms() - t_v);
  }
  hipError_t e = hipStreamSynchronize(stream);
  lock.lock();
  F.ending = false;
  F.timing.verify_ms = verify_ms;
  if (ensure_device(D) != 0) { F.busy = false; return -1; }
  if (verify && now != F.fp) {
    auto it = D.scene_cache.find(F.scene);
    if (it != D.scene_cache.end() && it->second->full_fp != now) {
      free_device_scene(it->second);        // (waits for the other lane if that renders from it)
      D.scene_cache.erase(it);
    }
    F.busy = false;
    F.d = nullptr;
    return render_frame_locked(F.scene, &F.image, F.p.samples, F.p.max_bounces, nullptr, nullptr, &F.camera, &F.p.seed);
  }
  F.busy = false;
  if (e != hipSuccess) return rt_fail("rt_frame_end: %s", hipGetErrorString(e));
  if (copy_image_out(&F.image, W.image, F.p.width, F.p.height, F.stream) != 0) return -1;
  HIP_TRY(hipEventRecord(W.ev_frame[4], F.stream));
  HIP_TRY(hipStreamSynchronize(F.stream));
  HIP_TRY(hipGetLastError());
  D.last_counters = F.d ? F.d->ls[1 + ticket].counters : nullptr;       // rt_get_counters() = this frame's
  F.d = nullptr;
  FrameTiming T = F.timing;
  T.gpu_prep_ms = event_ms(W.ev_frame[0], W.ev_frame[1]);
  T.gpu_path_ms = event_ms(W.ev_frame[1], W.ev_frame[2]);
  T.gpu_resolve_ms = event_ms(W.ev_frame[2], W.ev_frame[3]);
  T.gpu_copy_ms = event_ms(W.ev_frame[3], W.ev_frame[4]);
  T.total_ms = (float)(now_ms() - F.t_begin);
  D.timing = T;
  return 0;
}

// Where the time of the last frame behind render_thread_proc / render / rt_render_frame went (one-device frames; a
// multi-device frame reports total_ms only).  Host: stamp = the per-frame scene check, upload = scene upload when it
// happened, enqueue = launching the frame; GPU (HIP events on the frame's stream): prep = accumulator clear + the
// preparation kernel, path = the path kernel, resolve, copy = device-to-host copy of the image; total = wall clock of the call.
extern "C" int rt_get_frame_timing(RT_Frame_Timing *out) {
  if (!out) return rt_fail("rt_get_frame_timing: NULL");
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  out->stamp_ms = D.timing.stamp_ms; out->upload_ms = D.timing.upload_ms; out->enqueue_ms = D.timing.enqueue_ms;
  out->gpu_prep_ms = D.timing.gpu_prep_ms; out->gpu_path_ms = D.timing.gpu_path_ms; out->gpu_resolve_ms = D.timing.gpu_resolve_ms;
  out->gpu_copy_ms = D.timing.gpu_copy_ms; out->total_ms = D.timing.total_ms;
  out->verify_ms = D.timing.verify_ms; out->gather_ms = D.timing.gather_ms;
  out->n_devices = D.timing.n_devices; out->slowest_device = D.timing.slowest_device;
  return 0;
}

static int read_counters(Device &D, unsigned long long c[RT_N_COUNTERS]);
// Of the node visits of the last rt_render_accumulate launch: how many were COUNTED but not executed -- the one root visit of
// every camera path whose tile's pixel pyramid misses every child of the root (the reference, and the oracle, spend and count
// it; the kernel proves its outcome per tile and skips it).  bench.py's roofline carries it as a footnote.
extern "C" int rt_get_skipped_root_visits(u64 *out) {
  if (!out) return rt_fail("rt_get_skipped_root_visits: NULL");
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (ensure_device(D) != 0) return -1;
  unsigned long long c[RT_N_COUNTERS];
  if (read_counters(D, c) != 0) return -1;
  *out = c[7];
  return 0;
}

static int read_counters(Device &D, unsigned long long c[RT_N_COUNTERS]) {
  HIP_TRY(hipDeviceSynchronize());
  if (!D.last_counters) { memset(c, 0, RT_N_COUNTERS * sizeof(unsigned long long)); return 0; }
  HIP_TRY(hipMemcpy(c, D.last_counters, RT_N_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int rt_get_counters(RT_Counters *out) {
  if (!out) return rt_fail("rt_get_counters: NULL");
  {
    std::lock_guard<std::mutex> lk(g_multi_mutex);
    if (g_multi_counters_valid) { *out = g_multi_counters; return 0; }
  }
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (ensure_device(D) != 0) return -1;
  unsigned long long c[RT_N_COUNTERS];
  if (read_counters(D, c) != 0) return -1;
  out->paths = c[0];
  out->rays = c[1];
  out->node_visits = c[2];
  out->leaf_visits = c[3];
  out->shades = c[4];
  out->backgrounds = c[5];
  out->textured = c[6];
  return 0;
}

static float timed_slot_ms(Workspace &W, size_t slot) {
  if (hipEventSynchronize(W.ev1[slot]) != hipSuccess) return -1.0f;
  return event_ms(W.ev0[slot], W.ev1[slot]);
}

#ifdef RT_DIAG_VARIANTS
// Block statistics of the diagnostic kernel (RT_KERNEL=4): 8 pairs (executions, lanes) for
// shade, environment, regenerate, leaf-scalar, leaf-vector, node-scalar, node-vector, pop.
extern "C" int rt_get_sched_stats(u64 out[32]) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (ensure_device(D) != 0 || !out) return -1;
  unsigned long long c[RT_N_COUNTERS];
  if (read_counters(D, c) != 0) return -1;
  for (int i = 0; i < 32; i++) out[i] = c[8 + i];
  return 0;
}

// Block ledger of a -DRT_LEDGER build of the tile-stream kernel (LG_* slots, rt_dev.hip.h): out[0 .. n) = counters[8 .. 8 + n).
extern "C" int rt_get_ledger(u64 *out, i32 n) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (ensure_device(D) != 0 || !out || n < 0 || n > RT_N_COUNTERS - 8) return -1;
  unsigned long long c[RT_N_COUNTERS];
  if (read_counters(D, c) != 0) return -1;
  for (int i = 0; i < n; i++) out[i] = c[8 + i];
  return 0;
}

// Diagnostic kernel (RT_KERNEL=4): per wave start time, end time (100 MHz ticks) and items processed.
extern "C" int rt_get_wave_times(u64 *out, i32 max_waves) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (ensure_device(D) != 0 || !out || !D.ws.wave_times) return -1;
  int n = D.ws.wave_times_n < max_waves ? D.ws.wave_times_n : max_waves;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, D.ws.wave_times, (size_t)n * 3 * 8, hipMemcpyDeviceToHost));
  return n;
}
#endif  // RT_DIAG_VARIANTS

extern "C" f32 rt_last_kernel_ms(void) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (!D.ready || D.ws.n_timed == 0) return -1.0f;
  DeviceGuard guard(D);
  return timed_slot_ms(D.ws, (D.ws.n_timed - 1) % RT_MAX_TIMED);
}

extern "C" void rt_kernel_timing_reset(void) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  D.ws.n_timed = 0;
}

extern "C" f32 rt_kernel_timing_mean_ms(i32 *n_launches) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  size_t n = D.ws.n_timed < RT_MAX_TIMED ? D.ws.n_timed : RT_MAX_TIMED;
  if (n_launches) *n_launches = (i32)n;
  if (!D.ready || n == 0) return -1.0f;
  DeviceGuard guard(D);
  double sum = 0.0;
  for (size_t i = 0; i < n; i++) {
    float ms = timed_slot_ms(D.ws, i);
    if (ms < 0.0f) return -1.0f;
    sum += ms;
  }
  return (float)(sum / (double)n);
}

// ---------------------------------------------------------------------------------
// the reference's entry points

extern "C" void render_thread_proc(Rendering_Context *ctx) {
  if (!ctx) return;
  i32 c = __atomic_fetch_add(&ctx->_current_chunk, 1, __ATOMIC_SEQ_CST);
  if (c == 0) {
    // this entrant owns the frame (all rt_device_count() GPUs of it: render_frame_multi)
    int rc;
    {
      Device &D = dev0();
      std::lock_guard<std::mutex> lock(D.mutex);
      rc = render_frame_locked(ctx->scene, &ctx->image, ctx->samples, ctx->max_bounces, nullptr, nullptr);
    }
    (void)rc;   // failure text is in rt_last_error(); the context still completes
    i32 n_chunks = rt_chunk_count((i32)ctx->image.width, (i32)ctx->image.height);
    __atomic_store_n(&ctx->_current_chunk, n_chunks > 0 ? n_chunks : 1, __ATOMIC_SEQ_CST);
  }
  __atomic_fetch_add(&ctx->n_threads, -1, __ATOMIC_SEQ_CST);
}

extern "C" bool rendering_context_is_finished(Rendering_Context *context) {
  return __atomic_load_n(&context->n_threads, __ATOMIC_SEQ_CST) == 0;
}

extern "C" void rendering_context_finish(Rendering_Context *context) {
  while (__atomic_load_n(&context->n_threads, __ATOMIC_SEQ_CST) > 0) std::this_thread::yield();
}

// raytracer.c:722-784 on the GPU; semantics and the three documented choices
// (last triangle wins, texels outside the image skipped, per-texel seeding) are in oracle/oracle.h.
static int lightmap_bake_locked(Device &D, Image const *lightmap, Scene const *scene, isize samples) {
  if (ensure_device(D) != 0) return -1;
  if (!lightmap || !scene || !lightmap->pixels.data) return rt_fail("lightmap_bake: NULL argument");
  if (lightmap->pixel_type != PT_u8 || lightmap->components < 3) return rt_fail("lightmap_bake: need a u8 image with >= 3 components");
  if (samples <= 0 || lightmap->width <= 0 || lightmap->height <= 0 || lightmap->stride < lightmap->width)
    return rt_fail("lightmap_bake: bad size or sample count");
  RT_Device_Scene *d = cached_scene_locked(D, scene, nullptr, nullptr);
  if (!d) return -1;
  RT_KParams K;
  scene_only_kparams(&K, d);
  K.max_bounces = 8;            // cast_ray(scene, r, 8), raytracer.c:774
  K.seed = g_seed.load();
  const Triangles &T = scene->triangles;
  std::vector<float> verts((size_t)T.len * 9);
  for (int i = 0; i < T.len; i++)
    for (int k = 0; k < 3; k++) {
      verts[(size_t)i * 9 + 0 + k] = T.x[k][i];
      verts[(size_t)i * 9 + 3 + k] = T.y[k][i];
      verts[(size_t)i * 9 + 6 + k] = T.z[k][i];
    }
  size_t pb = (size_t)lightmap->stride * lightmap->height * lightmap->components;
  size_t ob = (size_t)lightmap->width * lightmap->height * sizeof(int);
  DevBuf b_verts, b_owner, b_pixels;
  HIP_TRY(b_verts.alloc(verts.size() * sizeof(float)));
  HIP_TRY(b_owner.alloc(ob));
  HIP_TRY(b_pixels.alloc(pb));
  float *dv = b_verts.as<float>();
  int *dow = b_owner.as<int>();
  uint8_t *dp = b_pixels.as<uint8_t>();
  HIP_TRY(hipMemcpy(dv, verts.data(), verts.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(dow, 0xFF, ob));                                        // owner = -1
  HIP_TRY(hipMemcpy(dp, lightmap->pixels.data, pb, hipMemcpyHostToDevice));   // untouched texels keep their value
  int rc = rt_launch_lightmap(&K, dv, T.len, (int)lightmap->width, (int)lightmap->height, (int)lightmap->stride,
                              (int)lightmap->components, (int)samples, dow, dp, nullptr);
  if (rc == 0) rc = (int)hipMemcpy(lightmap->pixels.data, dp, pb, hipMemcpyDeviceToHost);
  if (rc != 0) return rt_fail("lightmap_bake failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

extern "C" void lightmap_bake(Image const *lightmap, Scene const *scene, isize samples) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  {
    std::lock_guard<std::mutex> lk(g_multi_mutex);
    g_multi_counters_valid = false;
  }
  lightmap_bake_locked(D, lightmap, scene, samples);
}

extern "C" int render(Scene *scene, Image *image, isize samples, isize max_bounces) {
  return rt_render_frame(scene, image, samples, max_bounces, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------
// scene_init on the GPU (csrc/rt_build.hip): same Scene, byte for byte, as scene_init()

extern "C" int rt_gpu_build(const Triangle *h_tris, long n_in, long depth, BVH_Node *h_nodes, long n_internal, float *h_block,
                            long block_len, char *err, int err_len);

extern "C" int scene_init_gpu(Scene *scene, Triangle_Slice src, Allocator allocator) {
  if (!scene) return rt_fail("scene_init_gpu: scene is NULL");
  Device &D = dev0();
  {
    std::lock_guard<std::mutex> lock(D.mutex);
    if (ensure_device(D) != 0) return -1;
  }
  if (src.len < 0 || (src.len > 0 && !src.data)) return rt_fail("scene_init_gpu: bad triangle slice");
  if (src.len > (isize)1 << 27) return rt_fail("scene_init_gpu: %ld triangles are too many", (long)src.len);
  // The GPU build orders centroid keys with a radix sort of their bit patterns, which equals the `<` order of scene_init's
  // merge sort for every number including the infinities -- but not for NaN (`<` leaves a NaN where it stands, the radix
  // order puts it behind +inf).  A soup with a NaN coordinate is therefore built by scene_init itself: same Scene by definition.
  for (isize i = 0; i < src.len; i++)
    for (int v = 0; v < 3; v++) {
      const Vec3 &q = src.data[i].positions[v];
      if (q.x != q.x || q.y != q.y || q.z != q.z) {
        scene_init(scene, src, allocator);
        return 0;
      }
    }
  if (!rt_scene_alloc(scene, src.len, allocator)) return rt_fail("scene_init_gpu: the allocator failed");   // (drops a stale device copy)
  char err[256] = "";
  std::lock_guard<std::mutex> lock(D.mutex);
  DeviceGuard guard(D);
  int rc = rt_gpu_build(src.data, (long)src.len, (long)scene->bvh.depth, scene->bvh.nodes.data, (long)scene->bvh.nodes.len,
                        scene->triangles.x[0], (long)scene->triangles.len, err, (int)sizeof err);
  if (rc != 0) return rt_fail("scene_init_gpu: %s", err);
  return 0;
}

// ---------------------------------------------------------------------------------
// denoiser (reference denoiser.h / denoiser.c:131-153), SURVEY.md section 8f #3

extern "C" int rt_denoise(i32 width, i32 height, void const *d_src, void *d_dst, void *stream) {
  {
    Device &D = dev0();
    std::lock_guard<std::mutex> lock(D.mutex);
    if (ensure_device(D) != 0) return -1;
  }
  if (width <= 0 || height <= 0 || !d_src || !d_dst || d_src == d_dst) return rt_fail("rt_denoise: bad arguments");
  int rc = rt_launch_denoise(width, height, width, 3, width, 3, (const uint8_t *)d_src, (uint8_t *)d_dst, (hipStream_t)stream);
  if (rc != 0) return rt_fail("denoise kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

static int denoise_host(Image const *src, Image const *dst) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (ensure_device(D) != 0) return -1;
  if (!src || !dst || !src->pixels.data || !dst->pixels.data) return rt_fail("denoise_image: NULL image");
  if (src->pixels.data == dst->pixels.data) return rt_fail("denoise_image: src and dst must differ (denoiser.c:134)");
  if (src->width != dst->width || src->height != dst->height) return rt_fail("denoise_image: size mismatch");
  if (src->components < 1 || dst->components < 1 || src->stride < src->width || dst->stride < dst->width)
    return rt_fail("denoise_image: bad layout");
  size_t sb = (size_t)src->stride * src->height * src->components;
  size_t db = (size_t)dst->stride * dst->height * dst->components;
  DevBuf b_src, b_dst;
  HIP_TRY(b_src.alloc(sb));
  HIP_TRY(b_dst.alloc(db));
  uint8_t *ds = b_src.as<uint8_t>(), *dd = b_dst.as<uint8_t>();
  HIP_TRY(hipMemcpy(ds, src->pixels.data, sb, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dd, dst->pixels.data, db, hipMemcpyHostToDevice));     // components beyond 3 keep their values
  int rc = rt_launch_denoise((int)src->width, (int)src->height, (int)src->stride, (int)src->components,
                             (int)dst->stride, (int)dst->components, ds, dd, nullptr);
  if (rc == 0) rc = (int)hipMemcpy(dst->pixels.data, dd, db, hipMemcpyDeviceToHost);
  if (rc != 0) return rt_fail("denoise_image failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

extern "C" void denoise_image(Image const *src, Image const *dst, isize n_threads) {
  (void)n_threads;      // the reference's CPU thread count (denoiser.c:131); one kernel launch here
  denoise_host(src, dst);
}


#ifdef RT_DIAG_VARIANTS
// ---------------------------------------------------------------------------------
// unit-level device entry points (include/rt_hip_diag.h; diagnostic library only)

extern "C" int rt_test_math(i32 op, i32 n, f32 const *x, f32 const *y, f32 *out) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (ensure_device(D) != 0) return -1;
  if (n <= 0 || !x || !out) return rt_fail("rt_test_math: bad arguments");
  DevBuf bx, by, bout;
  size_t bytes = (size_t)n * sizeof(float);
  HIP_TRY(bx.alloc(bytes));
  HIP_TRY(bout.alloc(bytes));
  float *dx = bx.as<float>(), *dy = nullptr, *dout = bout.as<float>();
  HIP_TRY(hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice));
  if (y) {
    HIP_TRY(by.alloc(bytes));
    dy = by.as<float>();
    HIP_TRY(hipMemcpy(dy, y, bytes, hipMemcpyHostToDevice));
  }
  int rc = rt_launch_test_math(op, n, dx, dy, dout, nullptr);
  if (rc == 0) rc = (int)hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
  if (rc != 0) return rt_fail("rt_test_math failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

static int run_sweep(int (*launch)(unsigned long long *, hipStream_t), const char *name, u64 *out, int n_out) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (ensure_device(D) != 0) return -1;
  if (!out) return rt_fail("%s: NULL", name);
  DevBuf b;
  HIP_TRY(b.alloc((size_t)n_out * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(b.p, 0, (size_t)n_out * sizeof(unsigned long long)));
  int rc = launch(b.as<unsigned long long>(), nullptr);
  if (rc == 0) rc = (int)hipMemcpy(out, b.p, (size_t)n_out * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (rc != 0) return rt_fail("%s failed: %s", name, hipGetErrorString((hipError_t)rc));
  return 0;
}

// rcp_exact() (the six-instruction reciprocal of the leaf blocks) against the IEEE quotient over all 2^32 bit patterns:
// out[0] differing patterns inside its domain (must be 0), out[1] patterns outside the domain, out[2] differing ones
// among those, out[3] first differing pattern inside the domain + 1; out[4], out[5]: the same for rcp_leaf() (rt_hip_diag.h).
extern "C" int rt_test_rcp_sweep(u64 out[6]) { return run_sweep(rt_launch_test_rcp_sweep, "rt_test_rcp_sweep", out, 6); }

// The kernels' sRGB decode of a texture sample (division by 1.055 as a corrected multiplication) against
// rt_srgb_to_linear1() for every float in [0, 2] (and 4 M negative ones): out[0] patterns compared, out[1] differing (0 expected),
// out[2] first differing pattern + 1.
extern "C" int rt_test_srgb_sweep(u64 out[3]) { return run_sweep(rt_launch_test_srgb_sweep, "rt_test_srgb_sweep", out, 3); }

// The tile-stream kernel's shift-based fixed-point conversion of a sample against rt_accum_quantize() over all 2^32 bit
// patterns: out[0] differing patterns (0 expected), out[1] first differing pattern + 1.
extern "C" int rt_test_quantize_sweep(u64 out[2]) { return run_sweep(rt_launch_test_quantize_sweep, "rt_test_quantize_sweep", out, 2); }

// The tile order the preparation kernel derives from per-tile costs (rays of the previous launch): order[] must be a
// permutation of 0 .. n_tiles - 1 with non-increasing cost buckets (rt_kernels.hip: cost_bucket, 4 per power of two).
extern "C" int rt_test_tile_order(i32 n_tiles, u32 const *cost, u32 *order) {
  Device &D = dev0();
  std::lock_guard<std::mutex> lock(D.mutex);
  if (ensure_device(D) != 0) return -1;
  if (n_tiles <= 0 || !cost || !order) return rt_fail("rt_test_tile_order: bad arguments");
  DevBuf bc, bo, bn, bk, bw, bz;
  HIP_TRY(bc.alloc((size_t)n_tiles * 4));
  HIP_TRY(bo.alloc((size_t)n_tiles * 4));
  HIP_TRY(bn.alloc(((size_t)n_tiles + (size_t)(n_tiles + 63) / 64) * 4));
  HIP_TRY(bk.alloc(RT_N_COUNTERS * 8));
  HIP_TRY(bw.alloc(64));
  HIP_TRY(bz.alloc((size_t)n_tiles * 4));
  HIP_TRY(hipMemcpy(bc.p, cost, (size_t)n_tiles * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(bo.p, 0xFF, (size_t)n_tiles * 4));
  int rc = rt_launch_prepare(n_tiles, bn.as<uint32_t>(), bn.as<uint32_t>() + n_tiles, bk.as<unsigned long long>(), bw.as<uint32_t>(),
                             bz.as<uint32_t>(), bc.as<uint32_t>(), bo.as<uint32_t>(), nullptr);
  if (rc == 0) rc = (int)hipMemcpy(order, bo.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost);
  if (rc != 0) return rt_fail("rt_test_tile_order failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}


extern "C" int rt_test_trace(RT_Device_Scene *d, i32 n, f32 const *rays, f32 *out_t, i32 *out_tri, f32 *out_uv) {
  if (!d || n <= 0 || !rays || !out_t || !out_tri || !out_uv) return rt_fail("rt_test_trace: bad arguments");
  Device &D = *d->dev;
  std::lock_guard<std::mutex> lock(D.mutex);
  DeviceGuard guard(D);
  if (ensure_device(D) != 0) return -1;
  RT_KParams K;
  scene_only_kparams(&K, d);
  DevBuf br, bt, btri, buv;
  HIP_TRY(br.alloc((size_t)n * 24));
  HIP_TRY(bt.alloc((size_t)n * 4));
  HIP_TRY(btri.alloc((size_t)n * 4));
  HIP_TRY(buv.alloc((size_t)n * 8));
  float *dr = br.as<float>(), *dt = bt.as<float>(), *duv = buv.as<float>();
  int   *dtri = btri.as<int>();
  HIP_TRY(hipMemcpy(dr, rays, (size_t)n * 24, hipMemcpyHostToDevice));
  int rc = rt_launch_test_trace(&K, n, dr, dt, dtri, duv, nullptr);
  if (rc == 0) rc = (int)hipMemcpy(out_t, dt, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (rc == 0) rc = (int)hipMemcpy(out_tri, dtri, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (rc == 0) rc = (int)hipMemcpy(out_uv, duv, (size_t)n * 8, hipMemcpyDeviceToHost);
  if (rc != 0) return rt_fail("rt_test_trace failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

// n rays through traversal_blocks() -- the NODE / LEAF / pop code of the path kernels -- in the path kernel's launch geometry.
//   pyramid: NULL, or 19 floats (4 outward plane normals at [4 q .. 4 q + 2], the rays' common origin at [16 .. 18]): every
//            ray is then treated as a camera ray of one tile and node blocks take the pyramid-culled form where the path
//            kernel would; the caller guarantees that every ray starts at that origin and lies inside the four planes
//   exit_lanes: 1 .. 64, how many finished lanes end a round of blocks (the path kernel's `sched_thresh`, 48)
//   mode: 0 = the instance the path kernel would choose for this scene, 1 = force the IEEE division in the leaf blocks,
//         2 = nodes from L1 / L2 instead of the LDS copy
//   visits: [0] += ray_aabbs_hit_8 equivalents, [1] += ray_triangles_hit_8 equivalents
extern "C" int rt_test_trace_stream(RT_Device_Scene *d, i32 n, f32 const *rays, f32 const *pyramid, i32 exit_lanes, i32 mode,
                                    f32 *out_t, i32 *out_tri, f32 *out_uv, u64 visits[2]) {
  if (!d || n <= 0 || !rays || !out_t || !out_tri || !out_uv || !visits) return rt_fail("rt_test_trace_stream: bad arguments");
  if (exit_lanes < 1 || exit_lanes > 64) return rt_fail("rt_test_trace_stream: exit_lanes %d outside [1, 64]", exit_lanes);
  Device &D = *d->dev;
  std::lock_guard<std::mutex> lock(D.mutex);
  DeviceGuard guard(D);
  if (ensure_device(D) != 0) return -1;
  RT_KParams K;
  scene_only_kparams(&K, d);
  const int per_wave = (K.depth > 0 ? K.depth : 1) * 256 + 1536;
  int room = (160 * 1024 - 16 * per_wave) / 208;
  K.n_lds_nodes = d->n_nodes < room ? d->n_nodes : room;
  if (!d->boxes_ordered || mode == 2) K.n_lds_nodes = 0;
  K.pyr_nodes = K.n_lds_nodes;
  // the short reciprocal is valid while |det| < 2^102: edges <= 2^38 (as for frames) and, here, ray directions <= 2^16
  float dir_max = 0.0f;
  for (i32 i = 0; i < n; i++)
    for (int k = 3; k < 6; k++) {
      float m = fabsf(rays[(size_t)i * 6 + k]);
      if (!(m <= dir_max)) dir_max = m;
    }
  K.short_div = (mode != 1 && d->max_edge <= 0x1p38f && dir_max <= 0x1p16f) ? 1 : 0;
  const int smem = K.n_lds_nodes * 208 + 16 * per_wave;
  int n_blocks = (n + 16 * 64 * 4 - 1) / (16 * 64 * 4);                 // ~4 rays per lane
  if (n_blocks > D.num_cus) n_blocks = D.num_cus;
  if (n_blocks < 1) n_blocks = 1;
  DevBuf br, bp, bt, btri, buv, bv;
  HIP_TRY(br.alloc((size_t)n * 24));
  HIP_TRY(bp.alloc(19 * 4));
  HIP_TRY(bt.alloc((size_t)n * 4));
  HIP_TRY(btri.alloc((size_t)n * 4));
  HIP_TRY(buv.alloc((size_t)n * 8));
  HIP_TRY(bv.alloc(16));
  HIP_TRY(hipMemcpy(br.p, rays, (size_t)n * 24, hipMemcpyHostToDevice));
  if (pyramid) HIP_TRY(hipMemcpy(bp.p, pyramid, 19 * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(bv.p, 0, 16));
  int rc = rt_launch_test_trace_stream(&K, n, br.as<float>(), pyramid ? bp.as<float>() : nullptr, exit_lanes, n_blocks, smem,
                                       bt.as<float>(), btri.as<int>(), buv.as<float>(), bv.as<unsigned long long>(), nullptr);
  if (rc == 0) rc = (int)hipMemcpy(out_t, bt.p, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (rc == 0) rc = (int)hipMemcpy(out_tri, btri.p, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (rc == 0) rc = (int)hipMemcpy(out_uv, buv.p, (size_t)n * 8, hipMemcpyDeviceToHost);
  if (rc == 0) rc = (int)hipMemcpy(visits, bv.p, 16, hipMemcpyDeviceToHost);
  if (rc != 0) return rt_fail("rt_test_trace_stream failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

extern "C" int rt_test_texture(RT_Device_Scene *d, i32 tex, i32 n, f32 const *uv, f32 *out_rgb) {
  if (!d || n <= 0 || !uv || !out_rgb) return rt_fail("rt_test_texture: bad arguments");
  Device &D = *d->dev;
  std::lock_guard<std::mutex> lock(D.mutex);
  DeviceGuard guard(D);
  if (ensure_device(D) != 0) return -1;
  if (tex < 0) tex = d->bg_texture;
  if (tex >= d->n_textures) return rt_fail("rt_test_texture: texture %d of %d", tex, d->n_textures);
  RT_KParams K;
  scene_only_kparams(&K, d);
  DevBuf buv, bout;
  HIP_TRY(buv.alloc((size_t)n * 8));
  HIP_TRY(bout.alloc((size_t)n * 12));
  float *duv = buv.as<float>(), *dout = bout.as<float>();
  HIP_TRY(hipMemcpy(duv, uv, (size_t)n * 8, hipMemcpyHostToDevice));
  int rc = rt_launch_test_texture(&K, tex, n, duv, dout, nullptr);
  if (rc == 0) rc = (int)hipMemcpy(out_rgb, dout, (size_t)n * 12, hipMemcpyDeviceToHost);
  if (rc != 0) return rt_fail("rt_test_texture failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}
#endif  // RT_DIAG_VARIANTS
