#!/usr/bin/env python3
"""Convert a texture collection to what the C++ layer reads: binary PPM (P6) files + a list file.

    python tools/convert_textures.py <list.txt | image dir> <out_dir>

The input is the reference's `texture_dbases` list (one image path per line, TextureCollection,
DataGenerator.cpp:117-149) or a directory of images in any format Pillow decodes.  Writes
<out_dir>/tex%06d.ppm and <out_dir>/database.txt (with the trailing newline the reference's reader needs
for the last line, DG:124-126).  Python callers can skip this step: Generator.pool_from_list decodes directly."""
import os
import sys

from PIL import Image


def main():
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    src, out = sys.argv[1], sys.argv[2]
    if os.path.isdir(src):
        paths = sorted(os.path.join(src, f) for f in os.listdir(src) if not f.startswith("."))
    else:
        paths = [ln for ln in open(src).read().split("\n")[:-1] if ln.strip()]
    os.makedirs(out, exist_ok=True)
    written = []
    for k, p in enumerate(paths):
        try:
            img = Image.open(p).convert("RGB")
        except Exception as e:  # not an image: say so and go on
            print("skipped %s: %s" % (p, e), file=sys.stderr)
            continue
        dst = os.path.join(out, "tex%06d.ppm" % k)
        with open(dst, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % img.size)
            f.write(img.tobytes())
        written.append(os.path.abspath(dst))
    with open(os.path.join(out, "database.txt"), "w") as f:
        f.write("".join(w + "\n" for w in written))
    print("%d images -> %s" % (len(written), os.path.join(out, "database.txt")))


if __name__ == "__main__":
    main()
