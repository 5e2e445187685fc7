"""Moment operators of the HIP backend: density, velocity, momentum flux."""

from .macroscopic import Macroscopic as Macroscopic, ZeroMoment as ZeroMoment, FirstMoment as FirstMoment, SecondMoment as SecondMoment
