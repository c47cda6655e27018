"""Write N random JPEG/PNG images of assorted sizes into a folder (smoke input for --train_image_dataset / --train_image_path)."""
import os, sys
import numpy as np
from PIL import Image
out, n = sys.argv[1], int(sys.argv[2])
os.makedirs(os.path.join(out, "sub"), exist_ok=True)
rng = np.random.default_rng(0)
for i in range(n):
    h, w = int(rng.integers(700, 1500)), int(rng.integers(700, 1500))
    small = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
    img = Image.fromarray(small).resize((w, h), Image.BICUBIC)
    img.save(os.path.join(out, "sub" if i % 3 == 0 else "", f"{i:03d}." + ("png" if i % 4 == 0 else "jpg")))
print("wrote", n, "images to", out)
