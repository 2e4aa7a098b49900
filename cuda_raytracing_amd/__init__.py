"""MI355X ray-tracing core: Python plumbing over the in-tree C-ABI library (librt_amd.so).

`Renderer` mirrors the reference's Renderer (CR/Renderer.h:55-109) for scripts and tests; the
product path is the HIP library - nothing here computes pixels on the CPU.
"""
import os

from . import abi  # noqa: F401
from .host import ASSET_DIR, SCENE_DIR, HostScene, save_bmp  # noqa: F401


class DepthOfField:
    def __init__(self):
        self.enabled, self.focalDistance, self.aperture = False, 10.0, 10.0


class Settings:
    def __init__(self):
        self.depthOfField = DepthOfField()
        self.resampling_size = 1


class Renderer:
    """Renderer(width, height, textureWidth, textureHeight, scene): same call surface as the
    reference (CR/main.cpp:5-15) minus the window."""

    def __init__(self, width, height, texture_width, texture_height, scene, asset_dir=ASSET_DIR, device=0, mode=abi.RT_MODE_GPU_REF):
        from .device import DeviceScene
        self.width, self.height = width, height
        self.texture_width, self.texture_height = texture_width, texture_height
        self.settings = Settings()
        self.seed, self.mode, self.outputName = 1, mode, "output.bmp"
        self._device, self._asset_dir, self._DeviceScene = device, asset_dir, DeviceScene
        self.pixels, self.last_stats = None, None
        self.LoadScene(scene)

    def LoadScene(self, scene):
        path = scene if os.path.exists(scene) else os.path.join(SCENE_DIR, scene)
        self.host = HostScene(path, self._asset_dir)
        self.host.set_viewport(self.width, self.height)
        self.scene = self._DeviceScene(self.host, self._device)

    def _abi_settings(self):
        d = self.settings.depthOfField
        return abi.Settings(dof_enabled=int(d.enabled), spp=self.settings.resampling_size, focal_distance=d.focalDistance,
                            aperture=d.aperture, seed=self.seed, flags=self.mode)

    def Render(self):
        self.pixels, self.last_stats = self.scene.render_frame(self._abi_settings(), self.texture_width, self.texture_height,
                                                               cam=(self.width, self.height))
        return self.pixels

    def RenderInParallel(self, output, resampling, aperture, focalDistance):
        self.settings.resampling_size = resampling
        d = self.settings.depthOfField
        d.enabled, d.aperture, d.focalDistance = True, aperture, focalDistance
        self.outputName = output
        self.Render()
        save_bmp(output, self.pixels)
