// Headless C++ front end over the C ABI, source-compatible with the part of the reference's
// Renderer a program uses (CR/Renderer.h:55-109, CR/main.cpp:3-18):
//     Renderer r(1600, 900, 1920, 1080, "scene.txt");
//     r.settings.depthOfField.enabled = true;
//     r.RenderInParallel("out.bmp", 128, 150, 168);
// The window / OpenGL texture / ImGui panel of the reference are gone: frames are rendered into
// host memory and written as BMP.  Errors throw std::runtime_error (the reference exits).
#ifndef RT_RENDERER_HPP
#define RT_RENDERER_HPP
#include <cstdint>
#include <string>
#include <vector>

#include "rt_amd.h"

struct DepthOfField {  // CR/Scene.cuh:9-14
    bool enabled;
    double focalDistance;
    double aperture;
};
struct Settings {      // CR/Scene.cuh:16-20
    DepthOfField depthOfField;
    int resampling_size;
};

class Renderer {
public:
    // width/height: camera viewport (CR/Renderer_CUDA.cpp:467 passes them as camera W,H);
    // textureWidth/Height: size of the rendered image.
    // gpus > 1: frames are rendered by that many GPUs (0 .. gpus-1) through a scene group (rt_render_frame_multi); the
    // reference knows one GPU only (CR/Renderer.cpp:104).
    Renderer(int width, int height, int textureWidth, int textureHeight, const std::string& scene, const std::string& assetDir = "",
             int device = 0, int gpus = 1);
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void LoadScene(const std::string& scenePath);  // CR/Renderer.cpp:47-55
    // One full frame with the current `settings` (what one iteration of Renderer::Render shows,
    // CR/Renderer.cpp:493-564), kept in `pixels`.
    void Render();
    // CR/Renderer.cpp:566-583: set samples / aperture / focal distance, enable depth of field,
    // render the frame and save it as BMP.
    void RenderInParallel(const std::string& output, int resampling, double aperture, double focalDistance);
    // Progressive mode of CR/Renderer_CUDA.cpp:421-436: render the next of segmentation x segmentation
    // rectangles; returns false once the frame is complete.
    bool RenderNextSegment();
    void SaveTextureToBMP();  // CR/Renderer.cpp:156-169

    int width, height;
    int texture_width, texture_height;
    int segmentation = 10;
    Settings settings;
    std::string outputName = "output.bmp";
    std::string scenePath, assetDir;
    uint64_t seed = 1;
    uint32_t mode = RT_MODE_GPU_REF;   // semantics preset (RT_MODE_* / RT_F_*)
    std::vector<uint8_t> pixels;       // RGBA8, row 0 = bottom
    rt_stats lastStats{};

private:
    rt_settings abiSettings() const;
    rt_host_scene* host_ = nullptr;
    rt_scene* scene_ = nullptr;
    rt_scene_group* group_ = nullptr;  // gpus > 1 only
    int device_, gpus_;
    int x_progress = 0, y_progress = 0;
};
#endif
