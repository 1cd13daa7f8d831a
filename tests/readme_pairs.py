"""The reference's README sample table (README.md:72-83): (input image, output image, enhance arguments).
Both files of every pair are data fixtures copied from the reference's data/ directory into tests/golden/
(flower, BASELINE.json configs[0]) and tests/golden/readme/ (the others).  `mountain` is the row the README
keeps commented out (README.md:84); its files are in data/ all the same.

The two JPEG rows (paper.jpg progressive 4:2:0, rock2.jpg baseline 4:4:4) are read by host/jpeg.cpp, whose pixels are
bit-identical to libjpeg's default decode (tests/test_jpeg.py) -- the decoder behind the author's cv::imread.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

#  name            input                              output                                   nRow nCol  hx     hy   T   K   weights
PAIRS = [
    ("flower",        "flower-50.bmp",                 "flower-filtered.png",                 10, 20, 100.0, 30.0, 50, 30, [2, 3, 4, 1]),
    ("bird",          "readme/bird.bmp",               "readme/bird-filtered.png",            10, 20, 1000.0, 20.0, 10, 10, [1, 5, 5, 1]),
    ("canyon",        "readme/canyon-dawn-20.bmp",     "readme/canyon-filtered.png",          20, 10, 500.0, 30.0, 40, 10, [2, 7, 5, 1]),
    ("brickwall",     "readme/brickwall-20.bmp",       "readme/brickwall-filtered.png",       10, 20, 1000.0, 25.0, 30, 50, [2, 3, 3, 1]),
    ("conifer",       "readme/conifer-10.bmp",         "readme/conifer-filtered.png",         25, 15, 800.0, 20.0, 40, 100, [2, 3, 5, 1]),
    ("forest",        "readme/forest-10.bmp",          "readme/forest-filtered.png",          20, 10, 5000.0, 30.0, 10, 10, [4, 6, 6, 1.05]),
    ("snow-mountain", "readme/snow-mountain-15.bmp",   "readme/snow-mountain-filtered.png",   10, 20, 200.0, 30.0, 30, 10, [3, 10, 1, 1]),
    ("red-cherries",  "readme/red-cherries-10.bmp",    "readme/red-cherries-filtered.png",    20, 10, 400.0, 30.0, 50, 20, [2, 2, 2, 1]),
    ("mountain",      "readme/mountain-15.bmp",        "readme/mountain-filtered.png",        10, 20, 1000.0, 20.0, 50, 80, [2, 2, 2, 1]),
    ("paper",         "readme/paper.jpg",              "readme/paper-filtered.png",           20, 20, 1000.0, 40.0, 50, 20, [0.5, 1, 5, 1]),
    ("rock2",         "readme/rock2.jpg",              "readme/rock2-filtered.png",           20, 30, 500.0, 10.0, 50, 50, [4, 3, 4, 1]),
]


def cli_args(pair):
    """argv[3:] of `enhance` for a pair, formatted as the README writes them."""
    _, _, _, nr, nc, hx, hy, T, K, w = pair
    fmt = lambda v: ("%d" % v) if float(v) == int(v) else repr(float(v))
    return [str(nr), str(nc), fmt(hx), fmt(hy), str(T), str(K)] + [fmt(x) for x in w]


def paths(pair):
    return os.path.join(GOLDEN, pair[1]), os.path.join(GOLDEN, pair[2])
