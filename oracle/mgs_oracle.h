/*
 * mgs_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the VK3DGSR hot path of nvpro-samples/vk_gaussian_splatting
 * (reference @ /root/reference, VERSION 2026.1.6).  Every function cites the
 * reference file:line it follows.  Nothing in the product (vk_gaussian_splatting_amd/)
 * may include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Pinning status:
 *   - ingest (PLY/SPZ/.splat -> SplatSet, RDF->RUB, maxShDegree, flipSh): PINNED against
 *     the reference's own miniply/spz/splat_set.h compiled unmodified into oracle/_ref/
 *     (tests/golden/ingest_*.npz, generator tests/golden/make_golden.py).
 *   - everything that lives in Slang shaders or in Vulkan-dependent .cpp files
 *     (upload transform, keys/cull, sort order, projection, SH, raster, blend):
 *     "parity unpinned" — the reference ships no tests/golden vectors for it and
 *     cannot be built here (needs slangc, Vulkan, nvpro_core2, glm).
 */
#ifndef MGS_ORACLE_H
#define MGS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_FORMAT_FLOAT32 = 0, ORC_FORMAT_FLOAT16 = 1, ORC_FORMAT_UINT8 = 2 }; /* shaderio.h:60-62 */

/* ---- ingest-side semantics (src/splat_set.h) ---- */
int  orc_max_sh_degree(size_t f_rest_len, size_t splat_count);              /* splat_set.h:52-74   */
void orc_flip_sh_rdf_to_rub(float out15[15]);                               /* splat-types.h:55-83 */
void orc_convert_rdf_to_rub(float* positions, float* rotation, float* f_rest,
                            size_t splat_count, size_t coeffs_per_channel); /* splat_set.h:78-114  */

/* ---- load-time upload transform (src/splat_set_vk.cpp) ---- */
void orc_cov3d(const float* scale, const float* rotation, size_t n, float* cov6);                 /* :263-288 */
void orc_rgba(const float* f_dc, const float* opacity, size_t n, float* rgba);                    /* :313-345 */
void orc_sh_interleave(const float* f_rest, size_t n, int coeffs_per_channel, float* sh_out);     /* :356-435 */
int  orc_sh_stride(int coeffs_per_channel);
/* in-place quantise then dequantise exactly as the shaders read it back
 * (splat_set_vk.cpp:85-112 ; threedgs_particle_buffers.h.slang:72-90,112-207) */
void orc_quantize_roundtrip(float* data, size_t count, int format, int is_sh);
uint16_t orc_float_to_half(float f);
float    orc_half_to_float(uint16_t h);

/* ---- frame ---- */
typedef struct OrcFrame {
  float view[16];       /* glm column-major memory */
  float proj[16];       /* glm column-major, RH, clip z in [0,1]; proj[5] may be negative */
  float camera_pos[3];
  int   width, height;
  float splat_scale;          /* shaderio.h:261 */
  float frustum_dilation;     /* shaderio.h:264 */
  float alpha_cull_threshold; /* shaderio.h:265 */
  int   sh_degree;            /* shaderio.h:262 */
  int   front_to_back;        /* FRONT_TO_BACK macro: keys use +depth; blend uses "under" */
  int   frustum_culling;      /* 0 none, 1 at dist (default), 2 at raster (shaderio.h:84-86) */
  int   target_fp16;          /* 1: round the colour target to fp16 after every blend (RGBA16F) */
  int   ms_antialiasing;      /* MS_ANTIALIASING macro (threedgs.h.slang:63-76) */
  int   size_culling;         /* SIZE_CULLING_MODE (dist.comp.slang:93-134), default off (parameters.h:185) */
  float size_culling_min_pixels; /* shaderio.h:266 */
  int   debug_flags;          /* 1 POINT_CLOUD_MODE (threedgs.h.slang:108-110), 2 SHOW_SH_ONLY (mesh.slang:205-207),
                                 4 DISABLE_OPACITY_GAUSSIAN (frag.slang:248-255) */
  /* ---- 3DGUT pipeline only (PIPELINE_MESH_3DGUT; ignored by the 3DGS entry points) ---- */
  int   camera_model;         /* 0 CAMERA_PINHOLE, 1 CAMERA_FISHEYE (shaderio.h; threedgut_raster.mesh.slang:161-165) */
  int   extent_method;        /* 0 EXTENT_EIGEN, 1 EXTENT_CONIC (shaderio.h:96-97; default CONIC, parameters.h:190) */
  float fov_rad;              /* frameInfo.fovRad (gaussian_splatting.cpp:1168): fisheye focal + ray generation */
  float alpha_clamp;          /* shaderio.h:271, default 0.99 */
  float kernel_min_response;  /* KERNEL_MIN_RESPONSE, parameters.h:216, default 0.0113 */
  /* ---- stochastic paths ---- */
  int   stochastic;           /* STOCHASTIC_SPLAT (sortingMethod == SORTING_STOCHASTIC_SPLAT, shaderio.h:27): the render entry
                                 points keep a depth buffer (LESS_OR_EQUAL, depth write on: gaussian_splatting.cpp:1386-1396,
                                 2297-2298) and write accepted fragments opaque (frag.slang:265-290) */
  int   dof_mode;             /* 3DGUT: DOF_MODE, 0 DOF_DISABLED / 1 DOF_FIXED_FOCUS (shaderio.h:136-138) */
  float focus_dist, aperture; /* shaderio.h:278-279 */
  int   frame_sample_id;      /* shaderio.h:275 */
  int   kernel_degree;        /* 3DGUT: KERNEL_DEGREE (shaderio.h:112-119); 2 = quadratic (default, parameters.h:215) */
  int   pipeline_3dgut;       /* 1: the frame runs a 3DGUT pipeline.  Only consumer: frameInfo.focal as the DIST stage sees it —
                                 the fisheye focal for a fisheye camera on a 3DGUT pipeline, the pinhole focal otherwise
                                 (gaussian_splatting.cpp:1239-1251).  The orc_gut_* entry points imply 1. */
  int   normal_method;        /* NORMAL_METHOD (shaderio.h:126-128, gaussian_splatting.cpp:1678): 0 max-density plane, 1 iso-surface
                                 (ray / kernel-ellipsoid intersection).  Only consumer: the 3DGUT fragment's normal
                                 (threedgrt.h.slang:330-335); the 3DGS mesh shader always uses the max-density plane (mesh.slang:219) */
} OrcFrame;

/* atan2 as both the oracle and the kernels evaluate it in the fisheye dist-stage cull (dist.comp.slang:75-90): the reference's
 * is implementation-defined (SPIR-V Atan2), so a fixed unfused-fp32 polynomial stands in; y > 0 */
float orc_atan2_det(float y, float x);
/* projectPointFisheye validity of dist.comp.slang:78-86 for a view-space position (before the (1,1,-1) flip) */
int   orc_fisheye_cull_valid(const OrcFrame* f, const float view_pos[3]);

/* nvshaders/random.h.slang (nvpro_core2; NOT in the reference tree, fetched by its CMake): xxhash32(uint3), pcg, rand —
 * restated from the published file, unpinned (no vectors of it exist here). */
uint32_t orc_xxhash32(uint32_t x, uint32_t y, uint32_t z);
float    orc_rand(uint32_t* seed);
/* post.comp.slang:29-43: main = lerp(main, aux1, 1/(frame_sample_id+1)) over n floats */
void     orc_post_accumulate(float* main_image, const float* aux1, size_t n, int frame_sample_id);

typedef struct OrcInstance {
  const float* centers;  /* [count*3] */
  const float* cov6;     /* [count*6] */
  const float* rgba;     /* [count*4], already dequantised */
  const float* sh;       /* [count*sh_stride] interleaved [coef][rgb], may be NULL */
  const float* scales;   /* [count*3] log-space scales (scalesAddress), only read by size culling; may be NULL */
  uint32_t count;
  int      sh_degree;    /* of the splat set */
  int      sh_stride;    /* 0, 9, 24, 45 */
  float    transform[16];
  float    transform_inv[16];
  const float* rotations; /* [count*4] (w,x,y,z) as stored (rotationsAddress), only read by the surface normal; may be NULL */
} OrcInstance;

typedef struct OrcProjected {
  int   valid;          /* 0 => degenerate quad / discarded */
  float center_px[2];
  float ndc_z;
  float basis1[2], basis2[2]; /* pixels */
  float rgba[4];
  int   opacity_disabled;
} OrcProjected;

void orc_mat4_inverse(const float m[16], float out[16]);  /* glm::inverse restated (cofactors) */
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);

uint32_t orc_encode_key(float v);                                               /* dist.comp.slang:33-38 */
/* keys + ids of survivors in ascending global id; returns V.  dist.comp.slang:40-171 */
uint32_t orc_key_cull(const OrcFrame* f, const OrcInstance* inst, int n_inst,
                      uint32_t* keys, uint32_t* ids);
/* stable ascending LSD radix sort, 4x8 bit — vrdx vk_radix_sort.cc:262-416 */
void orc_sort_stable(uint32_t* keys, uint32_t* ids, uint32_t n);
/* per-splat raster front end — threedgs_raster.mesh.slang:111-291 */
void orc_project(const OrcFrame* f, const OrcInstance* inst, uint32_t local_idx, OrcProjected* out);
/* whole frame; rgba_out is [height][width][4] floats, row 0 = NDC y -1.
 * returns number of fragments blended.  stats[0]=V, stats[1]=valid quads */
uint64_t orc_render(const OrcFrame* f, const OrcInstance* inst, int n_inst,
                    float* rgba_out, uint64_t* stats);
/* same, but consuming an externally supplied draw order (global ids) */
uint64_t orc_render_order(const OrcFrame* f, const OrcInstance* inst, int n_inst,
                          const uint32_t* ids, uint32_t v, float* rgba_out, uint64_t* stats);

/* the same restricted to the inclusive pixel window win = {x0,y0,x1,y1}; rgba_out is the window's own
 * [y1-y0+1][x1-x0+1][4] buffer (== the crop of orc_render_order's image, bit for bit) */
uint64_t orc_render_window(const OrcFrame* f, const OrcInstance* inst, int n_inst,
                           const uint32_t* ids, uint32_t v, const int win[4], float* rgba_out);

/* ---- 3DGUT (unscented-transform projection + per-pixel particle response), SURVEY.md 8f rank 3 ----
 * per-splat front end: shaders/threedgut_raster.mesh.slang:111-254 with threedgut.h.slang:26-163 (sigma points,
 * GUT_* constants of threedgut_definitions.h.slang), threedgut_camera_projections.h.slang:84-239 (perfect pinhole /
 * fisheye of threedgut_camera_models.h.slang:65-136, global shutter), quaternions.h.slang:39-58.
 * per-fragment: shaders/threedgut_raster.frag.slang:87-183 with cameras.h.slang:27-82 (ray generation) and
 * threedgrt.h.slang:57-135,238-278 (canonical ray, quadratic kernel).
 * Not restated: rolling shutter (the reference marks it untested), depth of field and stochastic splats (their random
 * numbers come from nvshaders/random.h.slang of the absent nvpro_core2), kernel degrees other than the default 2.
 * The camera pose reaches the reference's projector as translation + glm::quat_cast(viewMatrix); rotating by that
 * quaternion is the view matrix product itself, which is what is evaluated here (glm is not in the reference tree). */
typedef struct OrcGutProjected {
  int   valid;
  float center_px[2];
  float ndc_z;
  float half1[2], half2[2];   /* half axes of the emitted quad in pixels: CONIC (ex,0),(0,ey); EIGEN basisVector1/2 */
  float rgba[4];              /* colour incl. SH; .a after the MS_ANTIALIASING compensation */
  float position[3];          /* model space */
  float scale[3];             /* exp(stored scale) */
  float inv_rot[9];           /* splatInvRotation rows (row-major): mul(v, inv_rot) == R(q)^T v */
} OrcGutProjected;
void     orc_project_gut(const OrcFrame* f, const OrcInstance* inst, uint32_t local_idx, OrcGutProjected* out);
/* one fragment: returns 1 and the opacity if the hit is accepted (threedgut_raster.frag.slang:87-127) */
int      orc_gut_fragment(const OrcFrame* f, const OrcInstance* inst, const OrcGutProjected* P, int px, int py, float* opacity);
/* the same, plus the fragment's world normal under NORMAL_METHOD_ISO_SURFACE (computeEllipsoidNormal, threedgrt.h.slang:423-497,
 * raySphereIntersection :502-540; to world space as particleProcessHitGutWithNormal :337-345); normal_world may be NULL */
int      orc_gut_fragment_iso(const OrcFrame* f, const OrcInstance* inst, const OrcGutProjected* P, int px, int py, float* opacity,
                              float thin_particle_threshold, float* normal_world);
/* whole frame in the supplied draw order (global ids); same blending / target semantics as orc_render_order */
uint64_t orc_render_gut_order(const OrcFrame* f, const OrcInstance* inst, int n_inst,
                              const uint32_t* ids, uint32_t v, float* rgba_out, uint64_t* stats);

/* octahedral normal coding, shaders/octahedral_normal.h.slang:27-87 */
uint32_t orc_oct_encode(const float n[3]);
void     orc_oct_decode(uint32_t packed, float out[3]);
/* world-space normal the mesh shader hands to the fragments of one splat (NEED_SURFACE_INFO):
 * threedgs_raster.mesh.slang:209-235, threedgrt.h.slang:42-48,358-419 (NORMAL_METHOD_MAX_DENSITY_PLANE),
 * quaternions.h.slang:33-76.  quantize != 0 applies the QUANTIZE_NORMALS round trip (frag.slang:200). */
void orc_splat_normal(const OrcFrame* f, const OrcInstance* inst, uint32_t local_idx, float thin_particle_threshold,
                      int quantize, float out[3]);
/* 3DGUT pipeline with NEED_SURFACE_INFO, front to back: picked depth, the splat that set it, integrated normal [4/pixel] (may be NULL) */
void     orc_render_surface_gut(const OrcFrame* f, const OrcInstance* inst, int n_inst, const uint32_t* ids_front_to_back, uint32_t v,
                                float depth_iso_threshold, float thin_particle_threshold, float* depth_out, uint32_t* id_out,
                                float* normal_out);

/* FTB surface side outputs, threedgs_raster.frag.slang:320-349 ; `ids` front-to-back.
 * depth_out/id_out: picked depth + the splat that set it.  normal_out (may be NULL): [H][W][4], the integrated
 * normal attachment = "under" blend of float4(normal * opacity, opacity) (gaussian_splatting.cpp:2090-2107),
 * kept in fp32 (the reference target is RGBA16F, gaussian_splatting.h:355). */
void orc_render_surface(const OrcFrame* f, const OrcInstance* inst, int n_inst, const uint32_t* ids, uint32_t v,
                        float depth_iso_threshold, float thin_particle_threshold, int quantize_normals,
                        float* depth_out, uint32_t* id_out, float* normal_out);

/* PSNR as the reference defines it: MSE over RGB / (W*H*3), 10*log10(1/MSE), cap 99.99
 * image_compare_metric.comp.slang:116-130, image_compare.cpp:869-893 */
double orc_psnr_rgb(const float* a, const float* b, int width, int height);

/* ---- CPU async sorter semantics — src/splat_sorter_async.cpp:92-141 ---- */
typedef struct OrcSortInstance {
  const float* positions; uint32_t count; uint32_t global_offset; float transform[16];
} OrcSortInstance;
/* fills distances[total], indices[total]; returns 0 on success.  threads<=0 => hardware_concurrency */
int orc_cpu_sort(const float dir[3], const float cop[3], const OrcSortInstance* inst, int n_inst,
                 uint32_t total, int front_to_back, int threads,
                 float* distances, uint32_t* indices, double* dist_ms, double* sort_ms);

#ifdef __cplusplus
}
#endif
#endif
