"""Car parameters of the ihm2 vehicle models.

Restates the values of the reference's ``python/constants.py:43-111`` (mass, geometry, drivetrain,
Pacejka base/derived coefficients, actuator time constants, aero and torque-vectoring gains).
``tests/test_constants.py`` checks every value against ``tests/golden/constants.json`` (dumped by
importing the reference module, see ``tests/golden/make_constants.py``).
"""
import math

g = 9.81
m = 230.0
I_z = 137.583
z_CG = 0.295
front_axle_track = rear_axle_track = axle_track = 1.24
l_R = 0.7853
l_F = 0.7853
wheelbase = 1.5706
rear_weight_distribution = l_R / wheelbase
front_weight_distribution = l_F / wheelbase
car_length = 3.19
car_width = 1.55

# drivetrain (python/constants.py:60-64)
C_m0 = 4.950
C_r0 = 297.030
C_r1 = 16.665
C_r2 = 0.6784

# Pacejka base parameters (python/constants.py:66-82)
b1s, b2s, b3s = -6.75e-6, 1.35e-1, 1.2e-3
c1s = 1.86
d1s, d2s = 1.12e-4, 1.57
e1s, e2s, e3s = -5.38e-6, 1.11e-2, -4.26
b1a, b2a = 3.79e1, 5.28e2
c1a = 1.57
d1a, d2a = -2.03e-4, 1.77
e1a, e2a = -2.24e-3, 1.81

# Pacejka constant-load version (python/constants.py:84-95)
static_weight = 0.5 * m * g * l_F / wheelbase
BCDs = (b1s * static_weight**2 + b2s * static_weight) * math.exp(-b3s * static_weight)
Cs = c1s
Ds = d1s * static_weight + d2s
Es = e1s * static_weight**2 + e2s * static_weight + e3s
Bs = BCDs / (Cs * Ds)
BCDa = b1a * math.sin(2 * math.atan(static_weight / b2a))
Ca = c1a
Da = d1a * static_weight + d2a
Ea = e1a * static_weight + e2a
Ba = BCDa / (Ca * Da)

# wheels (fdyn10 only, python/constants.py:97-101)
R_w = 0.20809
I_w = 0.3
k_d = 0.17
k_s = 15.0

# actuator time constants (python/constants.py:103-105)
t_T = 1e-3
t_delta = 0.02

C_downforce = 3.96864
K_tv = 300.0
marginal_constant = 0.01

NX = 8
NU = 2
NY = 12
NY_E = 8
NG = 2
NH = 2            # nonlinear track-boundary rows (old/generate_acaods_interface.py:191-212)
NC = 8 + NU + NG + NH   # two-sided constraint rows per stage: x boxes, u boxes, general rows, track rows
NLAM = 2 * NC     # one-sided multipliers per stage: NC lower sides, then NC upper sides
