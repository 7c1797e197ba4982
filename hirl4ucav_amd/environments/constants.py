"""Normalisation constants the observation uses (reference hirl/environments/constants.py:13-32; only Plane_position
and Plane_Euler_angles reach the observation, HarfangEnv_GYM.py:196-201,213-218)."""
import math

pi = math.pi

NormStates = {
    "Plane_position": 10000,
    "Plane_Euler_angles": pi,
}
