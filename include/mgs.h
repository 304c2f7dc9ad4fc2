/*
 * mgs.h — C ABI of the MI355X-native VK3DGSR hot path ("mgs" = MI355X gaussian splatting).
 *
 * This header is the drop-in boundary.  The reference
 * (nvpro-samples/vk_gaussian_splatting @ 2026.1.6) has no plugin/FFI API; the seam below is the
 * narrowest set of C++ member calls through which its renderer is reached.  Every entry point
 * cites the reference interface it replaces.  Plain pointers and sizes only; no exceptions
 * cross this boundary; every function returns an MgsStatus (0 = ok, <0 = error) and the
 * message is available from mgs_last_error() (thread-local).
 *
 * Matrix convention: float[16] in glm column-major memory order (m[col*4+row]), right-handed,
 * clip z in [0,1]; proj[5] may be negative (Vulkan Y flip) — exactly what
 * nvutils::CameraManipulator hands to GaussianSplatting::updateAndUploadFrameInfoUBO
 * (src/gaussian_splatting.cpp:1162-1173).
 *
 * Threading: handle-level thread compatibility — one thread per MgsScene at a time
 * (the reference has a single Vulkan submitter thread, src/gaussian_splatting.cpp:335).
 */
#ifndef MGS_H
#define MGS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGS_ABI_VERSION 4 /* 2: MgsFrameParams grew the 3DGUT fields; loader, strip-exchange and debug entry points added;
                            3: stochastic splats, depth of field, temporal accumulation (MgsFrameParams 256 -> 288 bytes);
                            4: MgsFrameOut 80 -> 88 bytes (escape_count) */

typedef enum MgsStatus {
  MGS_OK              = 0,
  MGS_ERR_INVALID_ARG = -1,
  MGS_ERR_IO          = -2,  /* file missing / unreadable            (PlyLoaderAsync E_FAILURE, ply_loader_async.h:37-44) */
  MGS_ERR_FORMAT      = -3,  /* not a valid 3DGS .ply/.spz/.splat    ("invalid 3DGS PLY file", ply_loader_async.cpp:449)  */
  MGS_ERR_DEVICE      = -4,  /* HIP error                            (NVVK_CHECK, never aborts here)                     */
  MGS_ERR_OOM         = -5,
  MGS_ERR_STATE       = -6,  /* call order violated (e.g. render before commit)                                          */
  MGS_ERR_OVERFLOW    = -7,  /* tile-pair capacity exceeded; frame is incomplete                                         */
  MGS_ERR_UNSUPPORTED = -8
} MgsStatus;

/* storage formats — shaders/shaderio.h:60-62 (FORMAT_FLOAT32/16/UINT8), chosen at
 * SplatSetVk::initDataStorage(shFormat, rgbaFormat) (src/splat_set_vk.h:112) */
enum { MGS_FORMAT_FLOAT32 = 0, MGS_FORMAT_FLOAT16 = 1, MGS_FORMAT_UINT8 = 2 };
/* sorting methods — shaders/shaderio.h SORTING_* / parameters.h:182 */
enum { MGS_SORT_GPU_RADIX = 0, MGS_SORT_CPU_ASYNC = 1,
       /* SORTING_STOCHASTIC_SPLAT (shaderio.h:27; threedgs_raster.frag.slang:265-290, threedgut_raster.frag.slang:150-172):
        * every fragment is accepted with probability alpha and written opaque, the depth test keeps the nearest accepted
        * one.  The reference skips the sort and lets the depth buffer resolve; here the sorted per-bin lists are walked
        * nearest first and the first accepted fragment ends the pixel (same winner, ties aside). */
       MGS_SORT_STOCHASTIC = 3 };
enum { MGS_DOF_DISABLED = 0, MGS_DOF_FIXED_FOCUS = 1 }; /* shaderio.h:136-138; DOF_AUTO_FOCUS picks the distance under the UI's
                                                           cursor and then behaves as FIXED_FOCUS: pass that distance */
/* frustum culling — shaders/shaderio.h:84-86 / parameters.h:184 */
enum { MGS_CULL_NONE = 0, MGS_CULL_AT_DIST = 1, MGS_CULL_AT_RASTER = 2 };
/* colour target — src/gaussian_splatting.h:338-340 (RGBA16F default, RGBA32F optional) */
/* colour target (gaussian_splatting.h:338-340, doc/overview: RGBA8 / RGBA16F (default) / RGBA32F); RGBA8 is linear UNORM */
enum { MGS_TARGET_RGBA16F = 0, MGS_TARGET_RGBA32F = 1, MGS_TARGET_RGBA8 = 2 };
/* visualisation modes: POINT_CLOUD_MODE (threedgs.h.slang:108-110), SHOW_SH_ONLY (mesh.slang:205-207),
 * DISABLE_OPACITY_GAUSSIAN (frag.slang:248-255) */
enum { MGS_DEBUG_POINT_CLOUD = 1, MGS_DEBUG_SH_ONLY = 2, MGS_DEBUG_OPACITY_GAUSSIAN_DISABLED = 4 };
/* alpha channel meaning: the reference's default back-to-front pipeline accumulates
 * A = sum(alpha) (src/gaussian_splatting.cpp:2083-2084); its FTB pipeline yields 1-T (:2071-2076). */
enum { MGS_ALPHA_COVERAGE = 0 /* 1-T */, MGS_ALPHA_SUM = 1 /* sum(alpha); disables early termination */ };
enum { MGS_NORMAL_MAX_DENSITY_PLANE = 0, MGS_NORMAL_ISO_SURFACE = 1 };
/* raster pipelines (parameters.h PIPELINE_MESH / PIPELINE_MESH_3DGUT), 3DGUT camera models and quad extents */
enum { MGS_PIPELINE_3DGS = 0, MGS_PIPELINE_3DGUT = 1 };
enum { MGS_CAMERA_PINHOLE = 0, MGS_CAMERA_FISHEYE = 1 };
enum { MGS_EXTENT_EIGEN = 0, MGS_EXTENT_CONIC = 1 };

typedef struct MgsSplatSet_t* MgsSplatSet; /* RAM model == struct SplatSet, src/splat_set.h:33-48 */
typedef struct MgsScene_t*    MgsScene;    /* device scene == SplatSetManagerVk + renderer buffers  */

/* The six SoA arrays of SplatSet (src/splat_set.h:36-42), INRIA semantics, ALREADY in the
 * renderer's RUB frame (i.e. after SplatSet::convertCoordinates(RDF,RUB), splat_set.h:78-114):
 * positions[3n], f_dc[3n], f_rest[f_rest_per_splat*n] channel-major (all R, all G, all B),
 * opacity[n] (logit), scale[3n] (log), rotation[4n] (w,x,y,z). */
typedef struct MgsSplatSetView {
  const float* positions;
  const float* f_dc;
  const float* f_rest;
  const float* opacity;
  const float* scale;
  const float* rotation;
  uint64_t     splat_count;
  uint32_t     f_rest_per_splat; /* 0, 9, 24 or 45 */
  int32_t      sh_degree;        /* out only: SplatSet::maxShDegree(), splat_set.h:52-74 */
} MgsSplatSetView;

/* ---- errors ---- */
const char* mgs_last_error(void);
const char* mgs_version(void);

/* ---- ingest: replaces PlyLoaderAsync::loadScene / innerLoad (src/ply_loader_async.h:62,
 * src/ply_loader_async.cpp:291-453).  Synchronous; dispatches on the lower-cased extension
 * (.ply via the INRIA property names, .spz = Niantic gzip "NGSP" v1-3, .splat = 32-byte records)
 * and leaves the set in RUB coordinates exactly like the reference. */
int  mgs_splatset_load(const char* path, MgsSplatSet* out);
/* copies the arrays; caller keeps ownership of the host pointers (valid only during the call) */
int  mgs_splatset_from_arrays(const MgsSplatSetView* view, MgsSplatSet* out);
/* borrow the set's arrays (valid until mgs_splatset_destroy) */
int  mgs_splatset_view(MgsSplatSet set, MgsSplatSetView* out);
void mgs_splatset_destroy(MgsSplatSet set);

/* ---- asynchronous ingest with a request queue: PlyLoaderAsync (src/ply_loader_async.h:35-106: one loader thread,
 * states E_READY / E_LOADING / E_LOADED / E_FAILURE, reset before the next load) plus the scene-load queue the UI drains
 * one file at a time (prmScene.sceneLoadQueue, src/gaussian_splatting_ui.cpp:1149-1152,1235-1436: files are queued, the
 * loader takes the next one when idle, a failure does not stop the queue).  Host only: no device is touched. */
typedef struct MgsLoader_t* MgsLoader;
enum { MGS_LOADER_READY = 1, MGS_LOADER_LOADING = 2, MGS_LOADER_LOADED = 3, MGS_LOADER_FAILURE = 4 };  /* ply_loader_async.h:37-44 */
int  mgs_loader_create(MgsLoader* out);                 /* PlyLoaderAsync::initialize: starts the loader thread */
void mgs_loader_destroy(MgsLoader loader);              /* shutdown: joins the thread, drops what is queued */
int  mgs_loader_push(MgsLoader loader, const char* path); /* pushLoadRequest: append to the queue (any state) */
/* getStatus: the state of the request at the head (READY when nothing is queued or loading); *queued = requests waiting
 * behind it, *path_out (optional, >= path_capacity bytes) = its file name */
int  mgs_loader_status(MgsLoader loader, int* state, uint32_t* queued, char* path_out, size_t path_capacity);
/* LOADED: hands over the splat set (caller owns it) and resets — the next queued file starts loading.
 * FAILURE: returns the load's error code (message in mgs_last_error) and resets likewise.  Other states: MGS_ERR_STATE. */
int  mgs_loader_take(MgsLoader loader, MgsSplatSet* out);

/* ---- scene: replaces SplatSetManagerVk::createSplatSet/createInstance/updateInstanceTransform/
 * processVramUpdates (src/splat_set_manager_vk.h:202,222-249,261).  `device` is the HIP ordinal. */
int  mgs_scene_create(int device, MgsScene* out);
void mgs_scene_destroy(MgsScene scene);
/* run on a caller-owned hipStream_t (NULL = the scene's own stream) */
int  mgs_scene_set_stream(MgsScene scene, void* hip_stream);
/* instances are concatenated in creation order into the global splat id space
 * (rebuildGlobalIndexTables, src/splat_set_manager_vk.cpp:2304-2360) */
int  mgs_instance_add(MgsScene scene, MgsSplatSet set, const float transform[16], int* instance_id);
int  mgs_instance_set_transform(MgsScene scene, int instance_id, const float transform[16]);
/* SplatSetVk::initDataStorage + initDataBuffers (src/splat_set_vk.cpp:117-170,188-480): builds the
 * device buffers (centres, 3D covariances, RGBA, interleaved SH) in the requested formats.
 * Idempotent; call again after changing formats (the reference's --updateData). */
int  mgs_scene_commit(MgsScene scene, int sh_format, int rgba_format);
uint64_t mgs_scene_splat_count(MgsScene scene); /* getTotalGlobalSplatCount, gaussian_splatting.cpp:369 */
/* ---- frame contexts: frames in flight over ONE resident scene.
 * The reference keeps a single copy of the splat buffers however many frames its application loop has in flight; that is why
 * processUpdateRequests waits for the device before it touches them (src/gaussian_splatting.cpp:1092-1111).  A frame context
 * is the per-frame-in-flight state of that loop: its own HIP stream, working buffers (slots, sort ping-pong, projected records,
 * bin lists, frame image, counters) and captured frame graphs, reading the scene's committed buffers.  The returned handle is
 * accepted by every frame-level entry point below (mgs_render, mgs_sort_keys, mgs_frame_*, mgs_timings_query, mgs_sync,
 * mgs_scene_set_stream, the multi-GPU exchange); the scene-editing entry points (mgs_instance_add, mgs_instance_set_transform,
 * mgs_scene_commit) return MGS_ERR_STATE on it — edit the scene.  Transforms set on the scene reach every context's next
 * frame; mgs_scene_commit waits for the frames in flight on all contexts, and each context re-sizes its working set on its
 * next frame.  Threading: one thread per handle at a time; do not edit or commit the scene while another thread renders one
 * of its contexts.  Contexts may outlive the scene handle (the data is freed with the last handle). */
int  mgs_frame_context_create(MgsScene scene, MgsScene* context_out);
void mgs_frame_context_destroy(MgsScene context);
/* capacity of this handle's per-bin splat lists in entries (4 B each); 0 restores the default of 32 per global splat.  Takes
 * effect before the next frame.  A frame whose lists do not fit is incomplete: mgs_frame_stats returns MGS_ERR_OVERFLOW and
 * sets bit 0 of error_flags (the analogue of the reference's fixed-size sorting buffers, splat_set_manager_vk.cpp:2304-2360,
 * which are sized for the splat count and cannot overflow because a quad is not a list entry). */
int  mgs_scene_set_list_capacity(MgsScene scene_or_context, uint64_t entries);
/* device bytes held by the committed scene data (shared by all its contexts) and by this handle's working set */
int  mgs_scene_memory_usage(MgsScene scene_or_context, uint64_t* scene_bytes, uint64_t* working_bytes);
/* The device keeps every splat set in a spatially coherent STORAGE ORDER (Morton order of the centres;
 * build-defined, the reference keeps file order).  Every id that crosses this ABI is in the caller's
 * order; this returns the permutation storage index -> caller's index of an instance's splat set.
 * Equal depth keys are drawn in storage order (the reference's tie order is nondeterministic). */
int  mgs_scene_storage_order(MgsScene scene, int instance_id, uint32_t* new_to_old, size_t count);
/* test/debug hook: copy a committed device buffer of a splat set to the host, dequantised to
 * fp32 exactly as the shaders would read it, in the caller's splat order.  which: 0 centres[3n] 1 cov[6n] 2 rgba[4n] 3 sh[stride*n] */
int  mgs_scene_download_set(MgsScene scene, int instance_id, int which, float* dst, size_t count);

/* ---- per-frame parameters: shaderio::FrameInfo (shaders/shaderio.h:238-317) as filled by
 * updateAndUploadFrameInfoUBO (src/gaussian_splatting.cpp:1150-1295) plus the raster knobs of
 * parameters.h:86-201.  focal and basisViewport are derived inside, exactly as :1218-1250. */
typedef struct MgsFrameParams {
  float   view[16];
  float   proj[16];
  float   camera_pos[3];
  int32_t width, height;
  float   splat_scale;          /* default 1.0    shaderio.h:261 */
  float   frustum_dilation;     /* default 0.2    shaderio.h:264 */
  float   alpha_cull_threshold; /* default 1/255  shaderio.h:265 */
  int32_t sh_degree;            /* default 3      shaderio.h:262 */
  int32_t sort_mode;            /* MGS_SORT_*      */
  int32_t frustum_culling;      /* MGS_CULL_*      (forced to AT_RASTER with CPU sort, gaussian_splatting_ui.cpp:1469-1479) */
  int32_t target_format;        /* MGS_TARGET_*    */
  int32_t alpha_mode;           /* MGS_ALPHA_*     */
  int32_t ms_antialiasing;      /* 0/1            threedgs.h.slang:63-76 */
  /* multi-GPU strip partition (no reference counterpart): this device renders
   * 16-pixel tile rows [strip_row_begin, strip_row_end); 0,0 = whole frame */
  int32_t strip_row_begin, strip_row_end;
  int32_t collect_timings;      /* 1: bracket each stage with hipEvents on the render stream and wait for them;
                                   2: record the events but do not wait (query later with mgs_timings_query) */
  int32_t cpu_sort_blocking;    /* CPU_ASYNC only: 1 = wait for the sorter (deterministic tests) */
  int32_t debug_flags;          /* MGS_DEBUG_* bits: the reference's visualisation modes (parameters.h:86-201) */
  int32_t size_culling;         /* 0/1, default 0 (parameters.h:185): drop splats whose projected extent is below ...  */
  float   size_culling_min_pixels; /* ... this many pixels, default 1.0 (shaderio.h:266, dist.comp.slang:93-134)      */
  int32_t surface_outputs;      /* 0/1, default 0: also produce the FTB side outputs of NEED_SURFACE_INFO
                                   (threedgs_raster.frag.slang:320-349; 3DGUT pipeline: threedgut_raster.frag.slang:195-228):
                                   picked depth + the splat that set it + the integrated normal */
  float   depth_iso_threshold;  /* default 0.7 (parameters.h:200): depth = ndc z of the first fragment after which
                                   the pixel's transmittance is below this */
  int32_t cpu_lazy_sort;        /* CPU_ASYNC only, default 1 (parameters.h:183): start a new sort only if the viewpoint changed */
  float   thin_particle_threshold; /* surface_outputs only, default 1e-6 (parameters.h:163): exp(scale) below this makes
                                   an axis degenerate for the splat normal (threedgrt.h.slang:358-419) */
  int32_t quantize_normals;     /* surface_outputs only, default 1 (parameters.h:195): the splat normal passes through
                                   the 2x16-bit octahedral code (octahedral_normal.h.slang) before it is integrated */
  /* ---- 3DGUT raster pipeline (PIPELINE_MESH_3DGUT): unscented-transform projection
   * (threedgut_raster.mesh.slang:111-254, threedgut.h.slang:26-163) + per-pixel particle response
   * (threedgut_raster.frag.slang:87-183, threedgrt.h.slang:57-135,238-278).  Keys, cull, sort and binning are shared. */
  int32_t pipeline;             /* MGS_PIPELINE_3DGS (default) | MGS_PIPELINE_3DGUT */
  int32_t camera_model;         /* 3DGUT: MGS_CAMERA_PINHOLE (default) | MGS_CAMERA_FISHEYE (perfect equidistant fisheye,
                                   threedgut_camera_models.h.slang:120-136; rays cameras.h.slang:46-82) */
  int32_t extent_method;        /* 3DGUT: MGS_EXTENT_CONIC (default, parameters.h:190) | MGS_EXTENT_EIGEN (shaderio.h:96-97) */
  float   fov_rad;              /* 3DGUT fisheye: frameInfo.fovRad (gaussian_splatting.cpp:1168,1243); 0 = derive the vertical
                                   field of view from proj[5] */
  float   alpha_clamp;          /* 3DGUT: default 0.99 (shaderio.h:271) */
  float   kernel_min_response;  /* 3DGUT: default 0.0113 (parameters.h:216) */
  /* ---- stochastic paths (ABI 3).  Random numbers: nvshaders/random.h.slang (xxhash32, pcg, rand) of nvpro_core2, which is
   * not part of the reference tree; restated from the published file (csrc/kernels_common.h). */
  int32_t dof_mode;             /* 3DGUT only: MGS_DOF_* — thin-lens perturbation of each pixel's ray, one sample per frame
                                   (threedgut_raster.frag.slang:104-109, cameras.h.slang:85-108) */
  float   focus_dist;           /* default 1.3   (shaderio.h:278) */
  float   aperture;             /* default 0.001 (shaderio.h:279) */
  int32_t frame_sample_id;      /* frameInfo.frameSampleId (shaderio.h:275): seeds the per-pixel random numbers of DoF and of
                                   MGS_SORT_STOCHASTIC; the caller counts it up while the view stands still
                                   (gaussian_splatting.cpp:3040-3075) */
  int32_t temporal_sampling;    /* 0/1 (post.comp.slang:29-43): the frame handed back is the running mean of the samples
                                   0..frame_sample_id of this scene (sample 0 restarts it: pass 0 whenever the view, the size
                                   or the scene changed, as updateFrameSampleId does); kept in fp32 */
  int32_t kernel_degree;        /* 3DGUT: KERNEL_DEGREE (shaderio.h:112-119, default 2 = quadratic, parameters.h:215): the generalised
                                   Gaussian of particleRayMaxKernelResponse (threedgrt.h.slang:83-127); 0,1,2,3,4,5,8 */
  int32_t normal_method;        /* 3DGUT + surface_outputs: NORMAL_METHOD (shaderio.h:126-128, parameters.h:121-125) — MGS_NORMAL_MAX_DENSITY_PLANE
                                   (default) | MGS_NORMAL_ISO_SURFACE: the fragment's normal is the normal of the kernel ellipsoid
                                   (3 sigma) where the pixel's ray enters it (threedgrt.h.slang:423-497).  The 3DGS pipeline's
                                   mesh shader always uses the max-density plane (threedgs_raster.mesh.slang:219). */
  int32_t reserved_[1];
} MgsFrameParams;

void mgs_frame_params_default(MgsFrameParams* p); /* fills the defaults cited above */

enum { MGS_STAGE_PROJECT = 0, MGS_STAGE_SORT = 1, MGS_STAGE_BIN = 2, MGS_STAGE_PAIRSORT = 3,
       MGS_STAGE_COMPOSITE = 4, MGS_STAGE_TOTAL = 5,
       MGS_STAGE_CULL = 6, /* the head of MGS_STAGE_PROJECT (included in it): from the frame's upload to the first kernel; the partition
                              cull ran here as a kernel of its own until round 3, it is part of the project kernels now */
       MGS_STAGE_COUNT = 8 };

typedef struct MgsFrameOut {
  void*    rgba_device;     /* device pointer: [height][width][4] fp16 (or fp32), row 0 = NDC y -1, linear */
  uint64_t rgba_bytes;
  uint32_t frustum_count;   /* survivors of the dist-stage cull == IndirectParams.instanceCount (shaderio.h:343-356) */
  uint32_t sorted_count;    /* elements actually sorted (after alpha/extent/off-screen rejection) */
  uint64_t tile_pairs;      /* (tile, splat) records built by the binning stage */
  uint32_t error_flags;     /* device-side diagnostics, 0 = clean.  bit 0: a per-bin list overflowed (mgs_frame_stats returns
                               MGS_ERR_OVERFLOW); bit 1: a bounded look-back wait of the key sort gave up — the sorted order,
                               hence the frame, is invalid (mgs_frame_stats / mgs_sort_keys / mgs_radix_sort_u32 return
                               MGS_ERR_DEVICE; nothing hangs) */
  uint32_t shaded_count;    /* (splat, screen region) pairs staged and shaded by the compositor (deferred SH evaluation) */
  uint64_t scanned_entries; /* bin-list entries the compositor looked at before its regions saturated */
  float    stage_ms[MGS_STAGE_COUNT]; /* valid when collect_timings; HIP-event times on the render stream */
  uint32_t escape_count;    /* sorted splats whose bin rectangle did not fit a code that rides through the key sort (more than 2 x 2
                               bins): the only ones whose rectangle is stored and gathered by id (build-only diagnostic) */
  uint32_t reserved0;
} MgsFrameOut;

/* GaussianSplatting::onRender -> renderHybridPipeline (src/gaussian_splatting.cpp:335,414,494):
 * processSortingOnGPU (:1298-1367) + drawSplatPrimitives (:1369-1465) + blending (:2066-2087).
 * Asynchronous on the scene's stream; counters in `out` are read back when the call returns
 * only if collect_timings or MGS_SYNC_STATS were requested via mgs_frame_stats(). */
int mgs_render(MgsScene scene, const MgsFrameParams* params, MgsFrameOut* out);
/* waits for the last mgs_render and fills counters / timings (readBackIndirectParametersIfNeeded, :1536) */
int mgs_frame_stats(MgsScene scene, MgsFrameOut* out);
/* per-stage HIP-event times of the timed frame rendered `frames_back` frames ago (0 = latest;
 * a ring of 128 timed frames is kept).  Waits only for that frame's last event. */
int mgs_timings_query(MgsScene scene, uint32_t frames_back, float stage_ms[MGS_STAGE_COUNT]);
/* side outputs of the last frame rendered with surface_outputs = 1, [height][width], row 0 = NDC y -1:
 * which 0: picked depth, float32 (0 where the transmittance never fell below the threshold);
 * which 1: global id (caller's order) of the splat that set it, uint32 (0xFFFFFFFF where none);
 * which 2: integrated normal, float32 x 4 per pixel = sum over the fragments (front to back) of
 *          (world normal * opacity, opacity) * transmittance — the RASTER_NORMAL attachment
 *          (gaussian_splatting.cpp:2090-2107, RGBA16F in the reference, fp32 here) */
int mgs_frame_download_surface(MgsScene scene, int which, void* host_dst, size_t bytes);
/* copy the last frame to the host (screenshot path, gaussian_splatting_ui.cpp:508-540, no tonemap) */
int mgs_frame_download(MgsScene scene, void* host_dst, size_t bytes);
/* copy this device's strip of the last frame into a caller-owned device buffer (all-gather staging) */
int mgs_frame_copy_strip(MgsScene scene, void* device_dst, size_t bytes);
/* ---- multi-GPU strip partition (no reference counterpart: the reference is single-GPU).  One process
 * per GPU, replicated splat buffers; every rank renders its tile rows of the SAME frame with the whole path, and the
 * strips are exchanged with RCCL (one grouped collective per frame on the scene's stream, in place in the frame buffer:
 * no staging copy).  Afterwards every rank's frame buffer holds the complete frame, bit-identical to a single-GPU
 * frame.  librccl is loaded on first use (dlopen); MGS_ERR_UNSUPPORTED when it is absent. */
#define MGS_COMM_ID_BYTES 128
/* ncclGetUniqueId: call on ONE rank, hand the 128 bytes to every rank out of band (MPI, torch.distributed, a file) */
int mgs_comm_unique_id(void* id_out);
/* ncclCommInitRank on the scene's device (collective: every rank calls it with the same id).  world_size 1 is valid. */
int mgs_scene_comm_init(MgsScene scene, int rank, int world_size, const void* id);
int mgs_scene_comm_destroy(MgsScene scene);
/* tile-row boundaries of all ranks, row_bounds[world_size + 1], ascending, [0] = 0, [world_size] >= tile rows of the
 * frame (16-pixel rows).  NULL restores the default: equal strips of ceil(rows / world_size).  Cost-balanced tables
 * come from mgs_frame_row_costs of earlier frames ("optionally cost-balanced from last frame's D"). */
int mgs_scene_set_strip_rows(MgsScene scene, const int32_t* row_bounds, int count);
/* mgs_render of this rank's strip + the exchange.  params->strip_row_* are ignored (the table decides).  A rank whose strip is
 * empty (equal consecutive bounds, more ranks than tile rows) renders nothing and only receives.  A rank whose own render fails
 * still joins the exchange (its rows are stale) and returns the render's error afterwards, so the peers never block; if it
 * cannot even hold a frame buffer it aborts its communicator, which fails the peers' collective instead of hanging it. */
int mgs_render_gathered(MgsScene scene, const MgsFrameParams* params, MgsFrameOut* out);
/* per 16-pixel tile row of the last full frame: bin-list entries attributed to that row (a bin's entries spread
 * evenly over its tile rows) — the cost proxy the strip balancing uses.  Waits for the frame.  Calibrate on full-frame
 * mgs_render calls: after mgs_render_gathered the lists cover this rank's rows only, every rank would derive a different
 * table and the exchange sizes would disagree — MGS_ERR_STATE. */
int mgs_frame_row_costs(MgsScene scene, uint32_t* cost_per_tile_row, size_t rows);

/* test/debug hook (no reference counterpart: these are the mesh shader's per-quad outputs,
 * threedgs_raster.mesh.slang:243-289, which the reference never stores): the projected records the last full frame
 * built for the given global splat ids (caller's id space; meaningful only for ids that frame sorted, see
 * mgs_sort_download).  out10[i] = { centre_px.x, centre_px.y, basisVector1.xy, basisVector2.xy (pixels),
 * opacity (after MS antialiasing), conservative half extents x/y of the visible footprint (pixels), 0 };
 * rect_out[i] (may be NULL) = the footprint's bin rectangle x0 | y0<<8 | x1<<16 | y1<<24. */
int mgs_frame_download_projected(MgsScene scene, const uint32_t* global_ids, size_t count, float* out10, uint32_t* rect_out);
int mgs_sync(MgsScene scene);

/* ---- sort only (metric hook): vrdxCmdSortKeyValueIndirect (3rdparty/vrdx/include/vk_radix_sort.h:73-78)
 * fed by dist.comp.slang, or SplatSorterAsync::sortAsync/consume (src/splat_sorter_async.h:84-128)
 * when params->sort_mode == MGS_SORT_CPU_ASYNC. */
typedef struct MgsSortOut {
  uint32_t count;      /* number of sorted elements (visible count) */
  float    key_ms;     /* dist stage */
  float    sort_ms;    /* radix sort (GPU) / std::sort (CPU) */
  float    hist_ms;    /* GPU: the up-front histogram kernel, included in sort_ms */
  uint32_t passes;     /* GPU: radix passes executed */
  uint32_t reserved[3]; /* [0] 1: pass 2 sorted on the rank of key >> 16 and pass 3 did not run; [1] occurring values of key >> 16 */
} MgsSortOut;
int mgs_sort_keys(MgsScene scene, const MgsFrameParams* params, MgsSortOut* out);
/* download the sorted keys (GPU mode: u32 encodeMinMaxFp32 keys; CPU mode: fp32 distances) and global ids.  `keys` may be
 * NULL.  In GPU mode the sorted keys exist after mgs_sort_keys only (a frame's last sort pass writes the ids alone, as the
 * raster stage reads nothing else): asking for them after mgs_render returns MGS_ERR_STATE. */
int mgs_sort_download(MgsScene scene, uint32_t* keys, uint32_t* ids, uint32_t capacity);

/* sort an arbitrary device-resident (key,value) u32 array with the frame's sort kernels (reduce-then-scan LSD radix)
 * (keys_device/values_device are overwritten with the result) — used by the sort parity tests
 * and the sorted-Gsplats/s microbenchmark. */
int mgs_radix_sort_u32(MgsScene scene, void* keys_device, void* values_device, uint32_t count,
                       int begin_bit, int end_bit, float* elapsed_ms);
/* host convenience for the above: uploads, sorts, downloads */
int mgs_radix_sort_host(MgsScene scene, uint32_t* keys, uint32_t* values, uint32_t count,
                        int begin_bit, int end_bit, float* elapsed_ms);

/* ---- camera helper (BUILD-DEFINED: nvutils::CameraManipulator lives in the absent nvpro_core2;
 * SURVEY.md §8c "parity unpinned").  Right-handed lookAt + perspective with clip z in [0,1];
 * flip_y != 0 negates proj[5] (Vulkan convention).  Defaults mirror src/camera_set.h:48-53. */
void mgs_camera_lookat_perspective(const float eye[3], const float center[3], const float up[3],
                                   float fov_y_degrees, float z_near, float z_far,
                                   int width, int height, int flip_y,
                                   float view_out[16], float proj_out[16]);
/* T*R*S instance transform, computeTransform (src/utilities.h:170-199); rotation = Euler degrees */
void mgs_compute_transform(const float scale[3], const float rotation_deg[3], const float translation[3],
                           float transform_out[16], float inverse_out[16]);

#ifdef __cplusplus
}
#endif
#endif /* MGS_H */
