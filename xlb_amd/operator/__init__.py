"""Operator base class of the HIP backend (registry + dispatch, see operator.py)."""

from .operator import Operator as Operator
