"""npy (H,W,3 float linear) -> 8-bit sRGB PNG, no dependencies beyond numpy/zlib."""
import sys, zlib, struct, numpy as np
def write_png(path, img):
    img = np.clip(img, 0, None); img = np.where(img <= 0.0031308, 12.92 * img, 1.055 * np.power(np.maximum(img, 1e-8), 1 / 2.4) - 0.055)
    a = (np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)
    h, w, _ = a.shape
    raw = b''.join(b'\x00' + a[y].tobytes() for y in range(h))
    def chunk(t, d): return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xffffffff)
    open(path, 'wb').write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))
if __name__ == '__main__':
    write_png(sys.argv[2], np.load(sys.argv[1]) * (float(sys.argv[3]) if len(sys.argv) > 3 else 1.0))
