"""Collision operators of the HIP backend: BGK, KBC, Smagorinsky LES BGK."""

from .collision import Collision as Collision, BGK as BGK, KBC as KBC, SmagorinskyLESBGK as SmagorinskyLESBGK
