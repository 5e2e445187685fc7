from .collision import Collision as Collision, BGK as BGK, KBC as KBC, SmagorinskyLESBGK as SmagorinskyLESBGK
