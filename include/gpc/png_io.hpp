// png_io.hpp -- minimal PNG codec over zlib for ndb::Buffer (host-side I/O, untimed).
//
// The reference links libpng (lib/gpc/buffer.hpp:197-474); its development headers are not
// part of this image, zlib's are.  This codec covers what the reference's readPNG accepts
// (8-bit gray, 8-bit RGB, 16-bit gray; non-interlaced) and what its writers emit (8-bit
// gray / RGB, filter type 0).  Anything else is reported like the reference reports it:
// a message on stdout and a non-zero return.
#ifndef GPC_PNG_IO_HPP
#define GPC_PNG_IO_HPP

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace ndb {
namespace pngio {

struct Image {
  int width = 0, height = 0;
  int color_type = 0;  // 0 gray, 2 RGB, 3 palette, 4 gray+alpha, 6 RGBA
  int bit_depth = 8;
  int channels = 1;
  std::vector<uint8_t> pixels;  // height * width * channels * (bit_depth/8), big-endian samples
};

inline uint32_t be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

inline int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// returns 0 on success; 1 not openable; 2 not a PNG; 3 unsupported / corrupt
inline int decode_file(const std::string& path, Image& img) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) return 1;
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, fp)) > 0) file.insert(file.end(), buf, buf + got);
  fclose(fp);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) return 2;
  size_t pos = 8;
  std::vector<uint8_t> idat;
  bool have_ihdr = false;
  int interlace = 0;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t* type = &file[pos + 4];
    if (pos + 12 + len > file.size()) return 3;
    const uint8_t* data = &file[pos + 8];
    if (!memcmp(type, "IHDR", 4) && len >= 13) {
      img.width = (int)be32(data);
      img.height = (int)be32(data + 4);
      img.bit_depth = data[8];
      img.color_type = data[9];
      interlace = data[12];
      have_ihdr = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + len;
  }
  if (!have_ihdr || img.width <= 0 || img.height <= 0) return 3;
  switch (img.color_type) {
    case 0: img.channels = 1; break;
    case 2: img.channels = 3; break;
    case 3: img.channels = 1; break;
    case 4: img.channels = 2; break;
    case 6: img.channels = 4; break;
    default: return 3;
  }
  if (interlace != 0 || (img.bit_depth != 8 && img.bit_depth != 16)) return 3;
  const size_t bpp = (size_t)img.channels * img.bit_depth / 8;
  const size_t stride = (size_t)img.width * bpp;
  // The header's size is checked against the DATA before anything is allocated for it: deflate expands at most 1032 : 1,
  // so an image the compressed stream cannot possibly fill (a forged or truncated file) is refused here instead of being
  // given gigabytes; 2^31 bytes of scanlines bounds what an honest one may ask for.
  if (img.width > (1 << 24) || img.height > (1 << 24)) return 3;
  const size_t need = (stride + 1) * (size_t)img.height;
  if (need > ((size_t)1 << 31) || need > (size_t)idat.size() * 1032 + 1024) return 3;
  std::vector<uint8_t> raw(need);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return 3;
  img.pixels.assign(stride * img.height, 0);
  for (int y = 0; y < img.height; ++y) {
    const uint8_t* in = &raw[(stride + 1) * y];
    uint8_t* cur = &img.pixels[stride * y];
    const uint8_t* up = y ? &img.pixels[stride * (y - 1)] : nullptr;
    const int ft = in[0];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0;
      const int b = up ? up[i] : 0;
      const int c = (up && i >= bpp) ? up[i - bpp] : 0;
      int pred;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: pred = paeth(a, b, c); break;
        default: return 3;
      }
      cur[i] = (uint8_t)(in[1 + i] + pred);
    }
  }
  return 0;
}

inline void put_chunk(FILE* fp, const char* type, const uint8_t* data, size_t len) {
  uint8_t hdr[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len,
                    (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
  fwrite(hdr, 1, 8, fp);
  if (len) fwrite(data, 1, len, fp);
  uLong crc = crc32(0L, hdr + 4, 4);
  if (len) crc = crc32(crc, data, (uInt)len);
  uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
  fwrite(c, 1, 4, fp);
}

// 8-bit gray (channels 1) or RGB (channels 3), rows tightly packed.  Returns 0 on success.
inline int encode_file(const std::string& path, const uint8_t* pixels, int width, int height, int channels) {
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) return 1;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  fwrite(sig, 1, 8, fp);
  uint8_t ihdr[13] = {(uint8_t)(width >> 24), (uint8_t)(width >> 16), (uint8_t)(width >> 8), (uint8_t)width,
                      (uint8_t)(height >> 24), (uint8_t)(height >> 16), (uint8_t)(height >> 8), (uint8_t)height,
                      8, (uint8_t)(channels == 3 ? 2 : 0), 0, 0, 0};
  put_chunk(fp, "IHDR", ihdr, 13);
  const size_t stride = (size_t)width * channels;
  std::vector<uint8_t> raw((stride + 1) * height);
  for (int y = 0; y < height; ++y) {
    raw[(stride + 1) * y] = 0;
    memcpy(&raw[(stride + 1) * y + 1], pixels + stride * y, stride);
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<uint8_t> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
    fclose(fp);
    return 3;
  }
  put_chunk(fp, "IDAT", comp.data(), clen);
  put_chunk(fp, "IEND", nullptr, 0);
  fclose(fp);
  return 0;
}

}  // namespace pngio
}  // namespace ndb
#endif
