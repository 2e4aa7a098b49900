/*
 * rt_amd.h — C ABI of the MI355X ray-tracing core.
 *
 * This is the boundary a host program binds instead of the reference's CUDA render path.
 * Every entry point cites the reference interface it replaces (paths relative to the
 * reference tree, CR/ = CUDA_RayTracing/).  Plain C: POD structs, pointers and sizes only.
 *
 * Conventions: every int-returning call gives 0 on success and a negative RT_ERR_* code on
 * failure (the reference prints and exit(1)s, CR/Renderer_CUDA.cpp:3-6; this library never
 * exits); rt_last_error() returns a thread-local message for the last failure.  The caller
 * owns every host buffer it passes; the library owns the device memory behind rt_scene.
 * Calls block until their result is complete and are not re-entrant per handle.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ---------------------------------------------------------------------- */
#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument / malformed description            */
#define RT_ERR_IO (-2)        /* file could not be opened or decoded             */
#define RT_ERR_DEVICE (-3)    /* HIP runtime error (message has the HIP string)  */
#define RT_ERR_NOGPU (-4)     /* no usable gfx950 device / extension not loaded  */
#define RT_ERR_LIMIT (-5)     /* a fixed capacity of the library was exceeded    */

/* ---- compile-time constants of the algorithm -------------------------------------------- */
#define RT_MAX_DEPTH 10        /* CR/Scene.cu:23  max_depth (CR/raytracer.cpp:14)          */
#define RT_SPEC_POWER 20       /* CR/Scene_Material.cu:3                                   */
#define RT_MAX_DREFL_DEP 2     /* CR/Scene_Material.cu:4                                   */
#define RT_EPS 1e-6            /* CR/vector3.cpp:11 and the literal 1e-6 all over the .cu  */
#define RT_BIG_DIST 1e100      /* CR/Primitive.cuh:105                                     */
#define RT_BEZIER_MAX_CTRL 10  /* CR/Primitive.cuh:82-83 fixed R[10], Z[10]                */

/* ---- primitive / light type tags (same order as CR/Primitive.cuh:20-29, CR/Light.cuh:8-13) */
enum {
    RT_PRIM_SPHERE = 0,
    RT_PRIM_PLANE = 1,
    RT_PRIM_SQUARE = 2,
    RT_PRIM_CYLINDER = 3,
    RT_PRIM_BEZIER = 4,
    RT_PRIM_TRIANGLE = 5,
    RT_PRIM_MESH = 6
};
enum { RT_LIGHT_POINT = 0, RT_LIGHT_SQUARE = 1, RT_LIGHT_SPHERE = 2 };

/* ---- semantics flags ---------------------------------------------------------------------
 * The reference has two renderers that disagree in small ways: the CPU path
 * (CR/raytracer.cpp, CR/primitive.cpp, CR/Light.cpp) and the CUDA path (the .cu files).  Each
 * disagreement is one bit; a cleared bit selects the CPU path's behaviour.               */
#define RT_F_NORMALIZE_PRIMARY 0x001u /* primary direction normalised before tracing (CR/RayTracer.cu:60,65,78) vs raw Camera::Emit (CR/camera.cpp:57-59) */
#define RT_F_SKY_Y 0x002u             /* sky gradient on dir.y (CR/Scene.cu:51) vs dir.z (CR/raytracer.cpp:169)                      */
#define RT_F_TIE_LATER_WINS 0x004u    /* closest hit accepts temp-best < 1e-6 (CR/Scene.cu:14) vs temp < best (CR/Scene.cpp:33)        */
#define RT_F_SHADOW_SELF_SKIP 0x008u  /* shadow tests skip the shaded primitive (CR/Shade.cu:38,67,97) vs test it (CR/Light.cpp:35,67)  */
#define RT_F_SQUARE_UV_NO_ORIGIN 0x010u /* square UV from C.Dx (CR/Primitive.cu:46-47) vs (C-O).Dx (CR/primitive.cpp:230-231)           */
#define RT_F_TEX_U8_TAPS 0x020u       /* the .cu texture fetch (CR/Primitive.cu:85-115): floorf, each bilinear tap truncated to 8 bits before
                                       * summing (helper_math.h:1842-1844), and taps (row U2, col V1) / (U1, V2) weighted rat_U (1 - rat_V) /
                                       * (1 - rat_U) rat_V because the texture is uploaded transposed (CR/Renderer_CUDA.cpp:149-152,
                                       * CR/Primitive.cu:109-110); vs double taps with the weights of CR/bmp.cpp:146-149 */
#define RT_F_PIXEL_MAP_TRUNC 0x040u   /* pixel->camera map int((double)x/tex_w*cam_w) (CR/RayTracer.cu:55-56) vs exact floor(x*cam_w/tex_w) */
#define RT_F_MESH_RAW_DIR 0x080u      /* mesh / triangle intersectors take the direction as given, so t is in units of |d| (CR/Collision_Mesh.cu:19-63, CR/Collision.cu:110-149; matters for shadow rays d = L-C and fuzzy reflections) vs normalised first like every other intersector */
#define RT_F_BEZIER_STACKED 0x100u    /* Bezier = stack of cylinders incl. the `closest` bookkeeping of CR/Collision_Bezier.cu:200-240 vs Newton on the true surface of revolution */
#define RT_F_STRICT_TRAVERSAL 0x200u  /* BVH walked exactly as CR/Collision_Mesh.cu:65-80: no t culling, no any-hit early out (results are identical either way; visit counts are not) */

#define RT_MODE_CPU_REF 0u
#define RT_MODE_GPU_REF                                                                          \
    (RT_F_NORMALIZE_PRIMARY | RT_F_SKY_Y | RT_F_TIE_LATER_WINS | RT_F_SHADOW_SELF_SKIP |           \
     RT_F_SQUARE_UV_NO_ORIGIN | RT_F_TEX_U8_TAPS | RT_F_PIXEL_MAP_TRUNC | RT_F_MESH_RAW_DIR |   \
     RT_F_BEZIER_STACKED)

/* ---- host-side scene description (flat POD; replaces the pointer graph of CR/Scene.cuh:22-32) */
typedef struct rt_vec3 {
    double x, y, z;
} rt_vec3;

/* CR/Primitive.cuh:7-16 Cuda_Material; texture is an index into rt_scene_desc.textures or -1 */
typedef struct rt_material {
    rt_vec3 color, absor;
    double refl, refr, diff, spec, rindex, drefl;
    int32_t texture;
    int32_t pad_;
} rt_material;

/* CR/Primitive.cuh:31-103.  One record per top-level primitive, tagged by `type`. */
typedef struct rt_primitive {
    int32_t type;     /* RT_PRIM_*                                          */
    int32_t is_light; /* isLightPrimitive (CR/Primitive.cuh:99)             */
    rt_material material;
    union {
        struct { rt_vec3 O, De, Dc; double R; } sphere;                 /* :37-41 */
        struct { rt_vec3 N, Dx, Dy; double R; } plane;                  /* :44-48 */
        struct { rt_vec3 O, Dx, Dy; } square;                           /* :51-55 */
        struct { rt_vec3 O1, O2; double R; } cylinder;                  /* :58-62 */
        struct {
            rt_vec3 O1, O2, N, Nx, Ny;
            double R_c;
            int32_t degree; /* (#P= lines) - 1, CR/primitive.cpp:272-279 */
            int32_t pad_;
            double R[RT_BEZIER_MAX_CTRL], Z[RT_BEZIER_MAX_CTRL];
        } bezier;                                                       /* :73-84 */
        struct { rt_vec3 O1, O2, O3, N; } triangle;                     /* :31-34 */
        struct { int32_t mesh; int32_t pad_; } mesh;                    /* index into rt_scene_desc.meshes */
    } u;
} rt_primitive;

/* CR/Light.cuh:15-25; light_primitive is an index into primitives or -1 (CR/Renderer_CUDA.cpp:266-296) */
typedef struct rt_light {
    int32_t type;
    int32_t light_primitive;
    rt_vec3 color, O, Dx, Dy;
    double R;
} rt_light;

/* CR/Camera.cuh:9-22; W,H are the viewport the renderer was built with (CR/Renderer_CUDA.cpp:17-18) */
typedef struct rt_camera {
    rt_vec3 O, N, Dx, Dy;
    int32_t W, H;
    double shade_quality, drefl_quality;
    double lens_W, lens_H;
} rt_camera;

/* 24-bit texture as decoded by CR/bmp.cpp:52-96: rgb[(row*width + col)*3 + {0,1,2}] = R,G,B,
 * row 0 = first row stored in the BMP file. */
typedef struct rt_texture {
    int32_t width, height;
    const uint8_t* rgb;
} rt_texture;

/* Flattened BVH of one mesh.  The tree is the one CR/primitive.cpp:552-616 builds (median
 * split on axis depth%3, one triangle per leaf).  Interior nodes and leaves live in two
 * arrays; a 32-bit "ref" addresses either: bit 31 set = leaf, low bits = triangle slot;
 * bit 31 clear = interior node index; RT_REF_END terminates.  A depth-first walk in the
 * reference's order needs no stack: at an interior node go to `left` if its box is hit,
 * else to `skip`; at a leaf go to its `skip`. */
#define RT_REF_LEAF 0x80000000u
#define RT_REF_END 0xFFFFFFFFu
typedef struct rt_bvh_node { /* 64 bytes, 64-byte aligned */
    double bmin[3], bmax[3];
    uint32_t left;     /* ref of the left child (the next item in pre-order)          */
    uint32_t skip;     /* ref of the first item after this subtree, or RT_REF_END     */
    uint32_t right;    /* ref of the right child (= skip of the left subtree)         */
    uint32_t preorder; /* index of this node in the reference's recursive visit order */
} rt_bvh_node;
typedef struct rt_bvh_leaf { /* 96 bytes: one triangle, stored in leaf pre-order */
    rt_vec3 O1, O2, O3;
    uint32_t skip;     /* ref of the next item in pre-order, or RT_REF_END            */
    uint32_t preorder; /* index of this leaf in the reference's recursive visit order */
    uint32_t face;     /* index of the face in the mesh file                          */
    uint32_t pad_;
    double pad2_;
} rt_bvh_leaf;
typedef struct rt_mesh {
    uint32_t root;       /* ref of the root (a leaf ref when the mesh has one triangle, RT_REF_END when empty) */
    uint32_t node_base;  /* first entry of this mesh in rt_scene_desc.nodes             */
    uint32_t node_count;
    uint32_t leaf_base;  /* first entry of this mesh in rt_scene_desc.leaves            */
    uint32_t leaf_count;
    uint32_t depth;      /* maximum depth (root = 0)                                    */
} rt_mesh;

typedef struct rt_scene_desc {
    rt_vec3 background_top, background_bottom; /* CR/Scene.cuh:24-25 */
    rt_camera camera;
    int32_t n_lights;
    int32_t n_primitives;
    int32_t n_textures;
    int32_t n_meshes;
    uint32_t n_nodes;
    uint32_t n_leaves;
    const rt_light* lights;
    const rt_primitive* primitives; /* in the reference's list order: light primitives first, then file order reversed (CR/raytracer.cpp:180-193,221-224) */
    const rt_texture* textures;
    const rt_mesh* meshes;
    const rt_bvh_node* nodes;   /* refs inside index these two arrays directly (all meshes share them) */
    const rt_bvh_leaf* leaves;
} rt_scene_desc;

/* CR/Scene.cuh:9-20 Settings + the knobs the reference hard-codes or lacks */
typedef struct rt_settings {
    int32_t dof_enabled;    /* depthOfField.enabled       */
    int32_t spp;            /* resampling_size            */
    double focal_distance;  /* depthOfField.focalDistance */
    double aperture;        /* depthOfField.aperture      */
    uint64_t seed;          /* key of the counter-based RNG that replaces curand (CR/RayTracer.cu:9-21) */
    uint32_t flags;         /* RT_F_* / RT_MODE_*         */
    uint32_t pad_;
} rt_settings;

/* Work counters of one render call.  Rays are counted the same way by the CPU oracle. */
typedef struct rt_stats {
    uint64_t rays_primary;   /* one per pixel sample                                   */
    uint64_t rays_secondary; /* reflect + refract rays that were intersected           */
    uint64_t rays_shadow;    /* one per light sample per shaded hit, traced or not (see rays_shadow_traced) */
    uint64_t nodes_visited;  /* interior boxes tested + leaves reached                 */
    uint64_t tris_tested;    /* leaf triangles tested                                  */
    uint64_t shaded_hits;    /* CalnDiffusion evaluations                              */
    uint64_t texture_fetches;/* bilinear lookups                                       */
    uint64_t pixels;
    double kernel_ms;        /* HIP-event time of the render kernel(s) on the render stream */
    double total_ms;         /* wall time of the call                                  */
    uint64_t rays_shadow_traced; /* of rays_shadow: those that were intersected with the scene.  The rest were
                                  * decided without a ray (lamp behind the surface, nothing that could stand
                                  * between point and lamp, point inside an umbra); the reference traces every
                                  * one, and so do the oracle and RT_F_STRICT_TRAVERSAL: there the two are equal */
} rt_stats;

/* ---- host scene model: replaces Raytracer::SetInput/CreateAll (CR/raytracer.cpp:195-275),
 * Mesh::LoadModel/BuildBVH (CR/primitive.cpp:379-471,531-616), Bmp::Input (CR/bmp.cpp:52-96),
 * Camera::Initialize (CR/camera.cpp:44-55).  Runs on the CPU; needs no GPU. ---------------- */
typedef struct rt_host_scene rt_host_scene;
/* Parse a scene file in the reference's text grammar.  Relative texture/mesh paths are
 * resolved against `asset_dir` (NULL = the process's working directory, as the reference). */
int rt_host_scene_load(const char* scene_path, const char* asset_dir, rt_host_scene** out);
/* Override camera W,H (the viewport size the reference passes at CR/Renderer_CUDA.cpp:467). */
int rt_host_scene_set_viewport(rt_host_scene* hs, int width, int height);
const rt_scene_desc* rt_host_scene_desc(const rt_host_scene* hs);
/* Non-fatal findings of the load (missing texture -> primitive left untextured, as the
 * reference's GPU path renders it: CR/Renderer_CUDA.cpp:143-163, CR/Scene_Material.cu:34). */
int rt_host_scene_warning_count(const rt_host_scene* hs);
const char* rt_host_scene_warning(const rt_host_scene* hs, int i);
void rt_host_scene_free(rt_host_scene* hs);
/* Write a 24-bit BMP the way BmpSave::SaveBMP does (CR/BmpSave.cpp:5-44): packed 54-byte
 * header, rows bottom-up, BGR.  Rows are padded to a multiple of 4 bytes as the format requires (the
 * reference writes them unpadded; the files are byte-identical whenever width*3 is a multiple of 4,
 * which covers every size the reference uses).  `rgba` is width*height*4 bytes, row 0 = bottom. */
int rt_bmp_save_rgba(const char* path, const uint8_t* rgba, int width, int height);

/* ---- device side ------------------------------------------------------------------------ */
typedef struct rt_scene rt_scene;

/* Select the GPU this thread renders on and check it is a gfx950 part.  Replaces
 * cudaSetDevice(0) (CR/Renderer.cpp:102-109). */
int rt_init(int device);
/* Render on `stream` (a hipStream_t cast to void*, NULL = the null stream) from now on. */
int rt_set_stream(void* hip_stream);

/* Who builds the four-wide trees the pruned traversal walks, from the calling thread's next rt_scene_create on
 * (the setting belongs to the thread, like the render stream):
 * 0 = the host (binned surface-area heuristic, the default),
 * 1 = the device, triangles in Morton order, cut top-down where the code changes (about 1 ms for 100,000
 *     triangles; frames 4-13 % slower than on the host's trees),
 * 2 = the device, the host's surface-area algorithm run one block per node (5 ms for 33,000 triangles, 15 ms
 *     for 103,000; frames within 1 % of those on the host's trees).
 * The image does not depend on the choice.  The reference builds its one tree on the CPU
 * (CR/primitive.cpp:552-616, kept as it is for parity) and has no device builder; without a call
 * the environment variable RT_AMD_TREE_BUILD=host|device|device_sah decides. */
int rt_set_tree_builder(int which);
/* Test support: reads the scene's four-wide trees back and checks them (every triangle of a mesh under its tree
 * exactly once, depth within the walk's stack, every child box around the exact bounds of its triangles).
 * *trees = number of trees, *cost = their surface-area measure (sum of child-box areas over the root-box area,
 * added up over the trees): lower = fewer box tests per ray.  RT_ERR_INVALID with a message if a tree is broken.
 * No counterpart in the reference. */
int rt_scene_check_trees(const rt_scene* scene, int* trees, double* cost);
/* Device time of the last scene's tree build in ms (0 when the host built the trees). */
double rt_scene_tree_build_ms(const rt_scene* scene);
/* The render kernel exists in two variants that differ in how shadow rays walk the meshes (each ray on
 * its own lane, or long walks shared out over idle lanes); they give the same image.  The choice is made
 * at rt_scene_create from the scene's contents: scenes with meshes get the shared variant (it is faster
 * wherever shadow rays walk meshes, and as fast elsewhere).  Returns the choice (0 = plain, 1 = shared);
 * the two times are always 0 (nothing is timed).  RT_AMD_SHARED_WALK=0|1 in the environment overrides
 * the rule (with 0 a scene whose meshes have four-wide trees renders with the general kernel: per-lane walks,
 * shading table in global memory).  No counterpart in the reference. */
int rt_scene_shared_walk(const rt_scene* scene, double* plain_ms, double* shared_ms);
/* The shared variant is built twice, for three and for four resident waves per SIMD (168 / 128 registers per lane).  Four pays
 * only where shadow rays walking meshes carry the frame; scenes with several meshes or more than 65536 triangles get it
 * (RT_AMD_WAVES=3|4 overrides).  Returns the build this scene's pruned renders use.  No counterpart in the reference. */
int rt_scene_waves_per_simd(const rt_scene* scene);
#ifdef RT_EXPERIMENTS
/* Only in libraries built with -DRT_EXPERIMENTS (A/B timing builds, tools/ab.sh), not in the shipped one.  Pruned renders have
 * a second form there: three lean kernels instead of one large one.  The paths are traced without their shadow rays (a record per
 * ray says what was hit and which shadow rays are still open), a second kernel traces all shadow rays, a third computes the
 * colours from the records; same image bit for bit, measured slower on every configuration.  It needs about 0.9 KB of device
 * memory per path of a launch (launches are cut at 2^25 paths) and at most 4 lamps.  Returns whether this scene's pruned renders
 * use it (RT_AMD_DEFER=0|1 in the environment at rt_scene_create decides) and, in *fallbacks, how many calls had to be rendered
 * again by the one kernel because a record buffer was too small.  No counterpart in the reference. */
int rt_scene_deferred_shadows(const rt_scene* scene, int* fallbacks);
#endif

/* Upload a scene: one slab, one copy.  Replaces createCudaSceneFromCPUScene
 * (CR/Renderer_CUDA.cpp:298-335). */
int rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
/* Replaces FreeCudaScene (CR/Renderer_CUDA.cpp:364-400). */
void rt_scene_destroy(rt_scene* scene);

/* Render the rectangle [x0,x0+w) x [y0,y0+h) of a tex_w x tex_h image into `rgba`
 * (tex_w*tex_h*4 bytes, device or host memory; pixels outside the rectangle are not
 * touched).  Same rectangle semantics as launchRayTraceKernel (CR/RayTracer.cuh:12,
 * CR/RayTracer.cu:97-118); cam_w/cam_h are its viewport_width/height.  `stats` may be NULL. */
int rt_render_tile(rt_scene* scene, const rt_settings* settings, int tex_w, int tex_h, int cam_w,
                   int cam_h, int x0, int y0, int w, int h, uint8_t* rgba, rt_stats* stats);

/* Render the tiles of a tex_w x tex_h image that belong to `part` of `n_parts` (image cut
 * into tile x tile squares, tile t owned by part t % n_parts) into `rgba`; with n_parts = 1
 * this is launchRayTraceKernelParallel (CR/RayTracer.cuh:13, CR/RayTracer.cu:134-169).
 * The image does not depend on the partition: the RNG is keyed by pixel and sample. */
int rt_render_frame(rt_scene* scene, const rt_settings* settings, int tex_w, int tex_h, int cam_w,
                    int cam_h, int tile, int part, int n_parts, uint8_t* rgba, rt_stats* stats);

/* rt_render_tile for callers that cannot allocate device memory: tight serving loops that must not touch the allocator,
 * harnesses that forbid hipMalloc.  No counterpart in the reference.  Same kernels and same result as rt_render_tile, but the
 * work buffers (queue heads, counters, the rectangle's list of 8x8 pixel blocks, the path colours of one launch) live in a fixed
 * 12.1 MiB array in device memory that every GPU the library runs on holds until the process ends; nothing is allocated or freed
 * by this call.  `rgba_device` must be device memory on the scene's GPU (tex_w*tex_h*4 bytes).  Limits (RT_ERR_LIMIT beyond
 * them): the rectangle may hold at most 16384 blocks of 8x8 pixels (about 1 M pixels) and settings->spp at most 8192; calls with
 * more than 2^19 paths run as several launches.  All calls on one GPU share that GPU's array: calls of one process that overlap
 * in time (other threads, other scenes) are serialised by a per-device lock; two processes have an array each.  Blocking, like
 * every other call: it copies the block list to the device, reads the counters back and synchronises the stream, so it cannot
 * be recorded into a HIP graph (it allocates nothing, which is all it promises). */
int rt_render_tile_noalloc(rt_scene* scene, const rt_settings* settings, int tex_w, int tex_h, int cam_w,
                           int cam_h, int x0, int y0, int w, int h, uint8_t* rgba_device, rt_stats* stats);

/* ---- several GPUs in one process --------------------------------------------------------------
 * launchRayTraceKernelParallel (CR/RayTracer.cuh:13, CR/RayTracer.cu:134-169) renders the whole frame
 * with one call on the one GPU the reference knows (cudaSetDevice(0), CR/Renderer.cpp:104).  A scene
 * group is that call for n GPUs: the scene is uploaded to each, the image is cut into tile x tile
 * squares, tile t (row-major) is rendered by member t % n, and the members' tiles are copied GPU to GPU
 * into the first member's frame.  The frame is bit-identical for every n. */
typedef struct rt_scene_group rt_scene_group;
int rt_device_count(void);
/* devices = NULL selects GPUs 0 .. n_gpus-1; otherwise devices[k] is the GPU of member k (a GPU may be
 * named more than once: its members then share it, which is only useful for testing).  Replaces
 * createCudaSceneFromCPUScene (CR/Renderer_CUDA.cpp:298-335) once per GPU. */
int rt_scene_group_create(const rt_scene_desc* desc, int n_gpus, const int* devices, rt_scene_group** out);
void rt_scene_group_destroy(rt_scene_group* group);
int rt_scene_group_size(const rt_scene_group* group);
/* Kernel time of one member in the last rt_render_frame_multi, in ms. */
double rt_scene_group_member_ms(const rt_scene_group* group, int member);
/* One frame on all GPUs of the group.  `rgba`: host memory, or device memory on the first member's GPU.
 * `stats` (may be NULL): counters summed over the members, kernel_ms = the slowest member's. */
int rt_render_frame_multi(rt_scene_group* group, const rt_settings* settings, int tex_w, int tex_h,
                          int cam_w, int cam_h, int tile, uint8_t* rgba, rt_stats* stats);
/* The copy kernels of that exchange on their own, for hosts that run one process per GPU and move the
 * packed tiles themselves (RCCL gather): pack the pixels of the tiles owned by `part` out of a frame
 * into a contiguous buffer (tiles in row-major order, each tile row by row), or write such a buffer
 * back into a frame.  Device pointers, on the calling thread's current GPU and stream.
 * rt_member_pixels gives the number of pixels (4 bytes each) of a part, -1 on bad arguments. */
long long rt_member_pixels(int tex_w, int tex_h, int tile, int part, int n_parts);
int rt_pack_tiles(const uint8_t* image_device, int tex_w, int tex_h, int tile, int part, int n_parts, uint8_t* packed_device);
int rt_unpack_tiles(const uint8_t* packed_device, int tex_w, int tex_h, int tile, int part, int n_parts, uint8_t* image_device);
/* rt_unpack_tiles for the packs of parts first_part .. n_parts-1 in one launch: the pack of part p starts at
 * packed_device + (p - first_part) * stride_bytes (what a gather into one buffer leaves on the receiving GPU). */
int rt_unpack_tiles_all(const uint8_t* packed_device, size_t stride_bytes, int tex_w, int tex_h, int tile, int first_part,
                        int n_parts, uint8_t* image_device);

/* Debug/parity: trace only the primary ray (sample 0, no jitter, no depth of field) of every
 * pixel of the rectangle.  hit_prim[i] = primitive index or -1, hit_t[i] = distance,
 * hit_leaf[i] = leaf pre-order index inside the hit mesh or -1.  If visit_log is not NULL,
 * the refs' pre-order indices of every BVH item visited by pixel i (all meshes, in
 * primitive order) are appended at visit_log[i*visit_cap ...] and visit_count[i] receives
 * the number visited (which may exceed visit_cap; only the first visit_cap are stored).
 * Outputs are host arrays of w*h entries, row-major in the rectangle. */
int rt_trace_primary(rt_scene* scene, const rt_settings* settings, int tex_w, int tex_h, int cam_w,
                     int cam_h, int x0, int y0, int w, int h, int32_t* hit_prim, double* hit_t,
                     int32_t* hit_leaf, uint32_t* visit_log, uint32_t* visit_count, int visit_cap);

const char* rt_last_error(void);
/* A short hash of the sources and compiler flags this library was built from ("unidentified" if the build script passed none).
 * Lets a measurement harness refuse profiler counters that were collected with another build.  No counterpart in the reference. */
const char* rt_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
