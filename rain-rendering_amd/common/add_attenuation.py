"""Fog-like rain attenuation pre-pass (reference common/add_attenuation.py:26-95; SURVEY 8f "next" #1).  It
produces `rainy_bg`, the main INPUT of the hot path, and runs on the device (csrc/rr_prepass.h); this class holds
the scalar constants of the model (Weber et al. 2015 as used by the reference) and the reference-signature call."""
import math

import numpy as np


class FogRain:
    def __init__(self, rain_intensity, focal, f_number, angle, exposure=2, camera_gain=20):
        self.rain_intensity = rain_intensity
        self.angle = angle
        self.focal = focal
        self.f_number = f_number
        self.exposure_time = exposure * 1e-3
        self.camera_gain = camera_gain

    def constants(self):
        """(beta_ext, beta_hg, irr_num, irr_den) for rr_prepass_in: the scalar part of fog_rain_layer,
        evaluated here so the device needs no pow/cos."""
        beta_ext = 0.312 * self.rain_intensity ** 0.67                              # :40-43
        g = 0.97
        cos_term = math.cos(math.radians(self.angle))
        beta_hg = (1 - (g ** 2)) / (4 * np.pi * ((1 + g ** 2 - 2 * g * cos_term) ** 1.5))   # :60-64
        return beta_ext, beta_hg, 4 * (self.f_number ** 2), self.exposure_time * self.camera_gain * np.pi   # :51-54

    def fog_rain_layer(self, image, depth, hip=None):
        """The reference's call (add_attenuation.py:26-95; generator.py:386) for one frame, on the device
        (rr_prepass_frames): image float BGR in [0,1], depth in metres (float32 or float64).  Generator.run does not
        come through here -- it chains the pre-pass and the streak path on the device without a host round trip."""
        from .. import hip_backend
        from . import imgops
        hip = hip or hip_backend.shared_context()
        hip.set_prepass_kernels(imgops.gaussian_kernel(25, 25), imgops.gaussian_kernel(15, 0))     # :79-80, bad_weather.py:815
        return hip.prepass_frames([dict(bg=image, depth=depth, fog=self.constants())], want_env=False)[0]['rainy_bg']
