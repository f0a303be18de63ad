"""Diagnostic for the friction KATs: per-env numbers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tests.test_physics_kat as K
np.set_printoptions(precision=4, suppress=True, linewidth=200)
n = 32
env, zero = K._settled_with_cylinder_along_x(n)
for _ in range(10): env.step(zero)
rq = K._get(env, "LT_F_ROOT_QUAT")
axis = np.array([K._quat_R(q)[:, 0] for q in rq])
vel = K._get(env, "LT_F_OBJ_LIN_VEL_W"); vel[:, :3] += 0.25 * axis; K._set(env, "LT_F_OBJ_LIN_VEL_W", vel)
r0 = K._rel_in_robot_frame(env)
mu = 0.5 * (K._get(env, "LT_F_ENV_PARAMS")[:, 1] + K._get(env, "LT_F_OBJ_PARAMS")[:, 3])
mass = K._get(env, "LT_F_OBJ_PARAMS")[:, 2]; ln = K._get(env, "LT_F_OBJ_PARAMS")[:, 1]
tr = []
for t in range(12):
    env.step(zero)
    r = K._rel_in_robot_frame(env)
    dv = (K._get(env, "LT_F_OBJ_LIN_VEL_W") - K._get(env, "LT_F_ROOT_LIN_VEL_W"))[:, :3]
    rq1 = K._get(env, "LT_F_ROOT_QUAT")
    va = np.array([K._quat_R(rq1[e])[:, 0] @ dv[e] for e in range(n)])
    tr.append((r[:, 0] - r0[:, 0], va))
print("mu     ", mu[:10]); print("mass   ", mass[:10]); print("len    ", ln[:10])
print("pred mm", 1e3 * 0.0625 / (2 * 9.81 * mu[:10]))
for t, (d, va) in enumerate(tr):
    print(f"t={t} slid mm", 1e3 * d[:10], " v_axis", va[:10])
# hold test outliers
env, zero = K._settled_with_cylinder_along_x(n)
for _ in range(25): env.step(zero)
r0 = K._rel_in_robot_frame(env)
for t in range(100):
    env.step(zero)
    if t in (10, 30, 60, 99):
        r = K._rel_in_robot_frame(env)
        print(f"hold t={t} dx mm", 1e3 * (r - r0)[:16, 0], "\n         dy mm", 1e3 * (r - r0)[:16, 1], "\n         z", r[:16, 2])
print("len", K._get(env, "LT_F_OBJ_PARAMS")[:16, 1], "rad", K._get(env, "LT_F_OBJ_PARAMS")[:16, 0])
