"""Fog-like rain attenuation pre-pass (reference common/add_attenuation.py:26-95).  It
produces `rainy_bg`, the main INPUT of the hot path; it runs on the host in numpy for now
(SURVEY 8f "next" #1).  Formulae: Weber et al. 2015 as used by the reference."""
import math

import numpy as np

from . import imgops


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

    def fog_rain_layer(self, image, depth):
        beta_ext = 0.312 * self.rain_intensity ** 0.67                              # :40-43
        f_ext = np.exp((-beta_ext) * (depth / 1000))                                 # :48 (depth in km)
        f_ext = np.tile(np.expand_dims(f_ext, axis=-1), (1, 1, 3))
        g = 0.97
        cos_term = math.cos(math.radians(self.angle))
        beta_hg = (1 - (g ** 2)) / (4 * np.pi * ((1 + g ** 2 - 2 * g * cos_term) ** 1.5))   # :60-64
        irradiance = (4 * (self.f_number ** 2) * image) / (self.exposure_time * self.camera_gain * np.pi)   # :51-54
        irradiance_mean = np.mean(irradiance.reshape(-1, 3), axis=0)
        l_in = np.clip(beta_hg * irradiance_mean * (1 - f_ext), 0, 1)                # :66-73
        f_ext = imgops.gaussian_blur(f_ext, 25, 25)                                  # :79-80
        l_in = imgops.gaussian_blur(l_in, 25, 25)
        return np.clip(image * f_ext + l_in, 0, 1)                                   # :85-86,93
