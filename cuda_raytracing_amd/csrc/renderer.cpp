#include "renderer.hpp"

#include <cstdio>
#include <stdexcept>

static void check(int rc, const char* what) {
    if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error());
}

Renderer::Renderer(int w, int h, int tw, int th, const std::string& scene, const std::string& assets, int device, int gpus)
    : width(w), height(h), texture_width(tw), texture_height(th), assetDir(assets), device_(device), gpus_(gpus < 1 ? 1 : gpus) {
    settings.resampling_size = 1;  // defaults of CR/Renderer.cpp:40-44
    settings.depthOfField.enabled = false;
    settings.depthOfField.aperture = 10.0;
    settings.depthOfField.focalDistance = 10.0;
    check(rt_init(device_), "rt_init");
    LoadScene(scene);
}

Renderer::~Renderer() {
    if (group_) rt_scene_group_destroy(group_);
    if (scene_) rt_scene_destroy(scene_);
    if (host_) rt_host_scene_free(host_);
}

void Renderer::LoadScene(const std::string& path) {
    rt_host_scene* hs = nullptr;
    check(rt_host_scene_load(path.c_str(), assetDir.empty() ? nullptr : assetDir.c_str(), &hs), "LoadScene");
    for (int i = 0; i < rt_host_scene_warning_count(hs); i++) std::fprintf(stderr, "warning: %s\n", rt_host_scene_warning(hs, i));
    rt_scene* ds = nullptr;
    rt_scene_group* grp = nullptr;
    int rc = rt_host_scene_set_viewport(hs, width, height);
    if (rc == RT_OK) rc = rt_scene_create(rt_host_scene_desc(hs), &ds);  // segments and single-GPU frames
    if (rc == RT_OK && gpus_ > 1) rc = rt_scene_group_create(rt_host_scene_desc(hs), gpus_, nullptr, &grp);
    if (rc != RT_OK) {
        if (ds) rt_scene_destroy(ds);
        rt_host_scene_free(hs);
        check(rc, "LoadScene");
    }
    if (group_) rt_scene_group_destroy(group_);
    group_ = grp;
    if (scene_) rt_scene_destroy(scene_);
    if (host_) rt_host_scene_free(host_);
    host_ = hs, scene_ = ds, scenePath = path;
    x_progress = y_progress = 0;
    pixels.assign((size_t)texture_width * texture_height * 4, 0);
}

rt_settings Renderer::abiSettings() const {
    rt_settings s{};
    s.dof_enabled = settings.depthOfField.enabled ? 1 : 0;
    s.spp = settings.resampling_size;
    s.focal_distance = settings.depthOfField.focalDistance;
    s.aperture = settings.depthOfField.aperture;
    s.seed = seed;
    s.flags = mode;
    return s;
}

void Renderer::Render() {
    rt_settings s = abiSettings();
    pixels.resize((size_t)texture_width * texture_height * 4);
    if (group_) check(rt_render_frame_multi(group_, &s, texture_width, texture_height, width, height, 32, pixels.data(), &lastStats), "Render");
    else check(rt_render_frame(scene_, &s, texture_width, texture_height, width, height, 32, 0, 1, pixels.data(), &lastStats), "Render");
}

bool Renderer::RenderNextSegment() {
    const int sw = texture_width / segmentation, sh = texture_height / segmentation;
    if (sw <= 0 || sh <= 0 || y_progress >= texture_height) return false;
    rt_settings s = abiSettings();
    const int w = x_progress + sw > texture_width ? texture_width - x_progress : sw;
    const int h = y_progress + sh > texture_height ? texture_height - y_progress : sh;
    check(rt_render_tile(scene_, &s, texture_width, texture_height, width, height, x_progress, y_progress, w, h, pixels.data(), &lastStats),
          "RenderNextSegment");
    x_progress += sw;
    if (x_progress >= texture_width) x_progress = 0, y_progress += sh;
    return y_progress < texture_height;
}

void Renderer::RenderInParallel(const std::string& output, int resampling, double aperture, double focalDistance) {
    settings.resampling_size = resampling;
    settings.depthOfField.enabled = true;
    settings.depthOfField.aperture = aperture;
    settings.depthOfField.focalDistance = focalDistance;
    outputName = output;
    Render();
    SaveTextureToBMP();
}

void Renderer::SaveTextureToBMP() { check(rt_bmp_save_rgba(outputName.c_str(), pixels.data(), texture_width, texture_height), "SaveTextureToBMP"); }
