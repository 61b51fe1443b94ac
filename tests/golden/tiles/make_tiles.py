#!/usr/bin/env python3
"""Writes tiles.obj / tiles.mtl / tiles_kd.ppm / tiles_bump.ppm / light.txt: a small textured scene authored for this
repository (the reference ships no assets): a 6 x 6 floor of quads with vt/vn records, and a pyramid with v/vt faces.
Exercises map_Kd / bump in the .mtl loader, the PPM decoder, GenerateLODs and the textured render types."""
import struct

lines = ["# tiles: authored for the textured-mode parity tests", "mtllib tiles.mtl", ""]
nv = 0
def v(x, y, z):
    global nv
    lines.append("v %g %g %g" % (x, y, z)); nv += 1
    return nv
nt = 0
def vt(u, w):
    global nt
    lines.append("vt %g %g" % (u, w)); nt += 1
    return nt
lines.append("vn 0 1 0")
G = 6
P = [[v(i * 2.0, 0.1 * ((i * 7 + j * 3) % 5), j * 2.0) for i in range(G + 1)] for j in range(G + 1)]
T = [[vt(i * 0.75, j * 0.75) for i in range(G + 1)] for j in range(G + 1)]
lines += ["", "usemtl floor"]
for j in range(G):
    for i in range(G):
        a, b, c, d = (P[j][i], T[j][i]), (P[j + 1][i], T[j + 1][i]), (P[j + 1][i + 1], T[j + 1][i + 1]), (P[j][i + 1], T[j][i + 1])
        lines.append("f %d/%d/1 %d/%d/1 %d/%d/1 %d/%d/1" % (a + b + c + d))
base = [v(4, 0.5, 4), v(8, 0.5, 4), v(8, 0.5, 8), v(4, 0.5, 8)]
apex = v(6, 4.5, 6)
t = [vt(0, 0), vt(2, 0), vt(1, 2)]
lines += ["", "usemtl stone"]
for k in range(4):
    lines.append("f %d/%d %d/%d %d/%d" % (base[(k + 1) % 4], t[0], base[k], t[1], apex, t[2]))
open("tiles.obj", "w").write("\n".join(lines) + "\n")
open("tiles.mtl", "w").write("""# materials of tiles.obj
newmtl floor
Ka 0.4 0.4 0.4
Kd 0.8 0.8 0.8
Ks 0.3 0.3 0.3
Ns 12
map_Kd tiles_kd.ppm
newmtl stone
Ka 0.3 0.3 0.35
Kd 0.6 0.6 0.7
Ks 0.5 0.5 0.5
Ns 20
map_Kd tiles_kd.ppm
bump tiles_bump.ppm
""")
open("light.txt", "w").write("2 9 -3\n")

def ppm(name, sx, sy, f):
    body = bytearray()
    for y in range(sy):
        for x in range(sx):
            body += bytes(f(x, y))
    open(name, "wb").write(b"P6\n# authored\n%d %d\n255\n" % (sx, sy) + bytes(body))

def kd(x, y):
    c = ((x // 4) + (y // 3)) & 1
    n = (x * 37 + y * 101) % 29
    return (200 + n, 60 + n, 40 + 2 * n) if c else (50 + n, 90 + 3 * n, 200 + n)
ppm("tiles_kd.ppm", 16, 12, kd)
ppm("tiles_bump.ppm", 8, 8, lambda x, y: ((x * x * 11 + y * 29 + x * y * 7) % 256,) * 3)
