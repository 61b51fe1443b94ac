#!/usr/bin/env python3
"""Writes cornell34.obj / cornell34.mtl / light.txt: a Cornell-box-scale scene authored for this repository (the
reference ships no assets): 5 walls + 2 rotated boxes = 17 quads = 34 triangles after fan triangulation.
Exercises the loader: quads, `v/vt`, `v//vn`, `v/vt/vn`, negative indices, mtllib/usemtl, comments, light.txt."""
import math

def box(cx, cz, w, d, h, ang):
    c, s = math.cos(ang), math.sin(ang)
    pts = []
    for y in (0.0, h):
        for dx, dz in ((-w, -d), (w, -d), (w, d), (-w, d)):
            pts.append((round(cx + c * dx - s * dz, 4), y, round(cz + s * dx + c * dz, 4)))
    return pts  # 0-3 bottom ring, 4-7 top ring

lines = ["# cornell34: authored for the MI355X LBVH/tracer parity tests", "mtllib cornell34.mtl", ""]
V = []
def v(p):
    V.append(p)
    lines.append("v %g %g %g" % p)
    return len(V)
# room [0,10]^3, open towards -z
room = [(0, 0, 0), (10, 0, 0), (10, 0, 10), (0, 0, 10), (0, 10, 0), (10, 10, 0), (10, 10, 10), (0, 10, 10)]
r = [v(p) for p in room]
lines += ["vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1", "vn 0 1 0", "vn 0 0 -1", ""]
lines += ["usemtl white", "f %d/1/1 %d/2/1 %d/3/1 %d/4/1" % (r[0], r[1], r[2], r[3]),          # floor, v/vt/vn
          "f %d %d %d %d" % (r[4], r[7], r[6], r[5]),                                           # ceiling, v only
          "f %d//2 %d//2 %d//2 %d//2" % (r[3], r[2], r[6], r[7]),                                # back wall, v//vn
          "usemtl red", "f %d/1 %d/2 %d/3 %d/4" % (r[0], r[3], r[7], r[4]),                      # left wall, v/vt
          "usemtl green", "f -7 -3 -2 -6", ""]                                                   # right wall, negative indices
for name, (cx, cz, w, d, h, ang) in (("short", (3.2, 3.4, 1.5, 1.5, 3.0, 0.30)), ("tall", (6.6, 6.3, 1.5, 1.5, 6.0, -0.35))):
    b = [v(p) for p in box(cx, cz, w, d, h, ang)]
    lines.append("usemtl " + name)
    for q in ((4, 5, 6, 7), (0, 3, 2, 1), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)):
        lines.append("f %d %d %d %d" % tuple(b[i] for i in q))
    lines.append("")
open("cornell34.obj", "w").write("\n".join(lines) + "\n")
open("cornell34.mtl", "w").write("""# materials of cornell34.obj
newmtl white
Ka 0.4 0.4 0.4
Kd 0.75 0.75 0.75
Ks 0.1 0.1 0.1
Ns 8
newmtl red
Ka 0.3 0.05 0.05
Kd 0.65 0.06 0.05
Ks 0.1
Ns 8
newmtl green
Ka 0.05 0.3 0.05
Kd 0.12 0.45 0.15
Ks 0.1
Ns 8
newmtl short
Ka 0.3 0.3 0.2
Kd 0.7 0.7 0.5
Ks 0.4 0.4 0.4
Ns 24
newmtl tall
Ka 0.2 0.25 0.35
Kd 0.4 0.5 0.8
Ks 0.6 0.6 0.6
Ns 40
""")
open("light.txt", "w").write("5 9.5 4\n")
