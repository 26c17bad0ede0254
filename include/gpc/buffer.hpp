// buffer.hpp -- ndb:: containers and PODs of the openGPC API, without Eigen/libpng.
//
// Host-side mirror of the reference's lib/gpc/buffer.hpp for the types the inference path
// exposes: Point / Descriptor / Support / Correspondence (buffer.hpp:52-102), RGBColor
// (:41-50), Buffer<T> (:142-193: row-major, columns padded to a multiple of 16, public
// width/height), clearBoundary (:630-654), PNG read/write (:197-474) and the KITTI-colour
// disparity overlay (:949-1014).  Same names, argument meaning and error behaviour
// (messages on stdout, sentinel return values); storage is a vector, zero-initialised by the public constructors (the
// reference's Eigen arrays are not initialised at all; Buffer::uninitialized gives the library's own outputs the same).
#ifndef GPC_AMD_NDB_BUFFER_HPP
#define GPC_AMD_NDB_BUFFER_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "gpc/png_io.hpp"

namespace ndb {

struct RGBColor {
  uint8_t b, g, r;
  RGBColor(uint8_t r_, uint8_t g_, uint8_t b_) : b(b_), g(g_), r(r_) {}
  RGBColor() : b(0), g(0), r(0) {}
};

struct Point {
  int x, y;
  Point(int x_, int y_) : x(x_), y(y_) {}
  Point() : x(0), y(0) {}
};

// 24 bytes, like the reference's (buffer.hpp:58-62)
struct Descriptor {
  Point point;
  uint64_t state = 0;
  bool srcDescr = false;
  Descriptor(Point p, uint64_t s) : point(p), state(s) {}
  Descriptor() {}
  bool operator==(const Descriptor& d) const { return state == d.state; }
  bool operator!=(const Descriptor& d) const { return state != d.state; }
  bool operator<(const Descriptor& d) const { return state < d.state; }
  bool operator<=(const Descriptor& d) const { return state <= d.state; }
  bool diffImgs(const Descriptor& d) const { return srcDescr != d.srcDescr; }
  int operator%(const int& d) const { return (int)(state % (uint64_t)d); }
};

// layout == gpc_support of the C ABI
struct Support {
  int x, y;
  float d;
  Support(int x_, int y_, float d_) : x(x_), y(y_), d(d_) {}
  Support(int x_, int y_) : x(x_), y(y_), d(0.f) {}
  Support() : x(0), y(0), d(0.f) {}
};

// layout == gpc_correspondence of the C ABI
struct Correspondence {
  Point srcPt, tarPt;
  Correspondence(Point s, Point t) : srcPt(s), tarPt(t) {}
  Correspondence() {}
};

struct Dimension {
  int w, h;
  Dimension(int w_, int h_) : w(w_), h(h_) {}
};

inline int align16(int x) { return (x % 16) == 0 ? x : ((x / 16) + 1) * 16; }

namespace detail {
// std::allocator whose value-less construct() default-initialises: resize(n) of a vector of bytes touches no memory, so an
// image the device is about to fill is written once, not twice
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U>
  struct rebind {
    typedef default_init_allocator<U> other;
  };
  default_init_allocator() {}
  template <class U>
  default_init_allocator(const default_init_allocator<U>&) {}
  template <class U>
  void construct(U* p) {
    ::new (static_cast<void*>(p)) U;
  }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
  }
};
}  // namespace detail

template <class T>
class Buffer {
 public:
  int width = 0;   // visible width  (<= cols())
  int height = 0;  // visible height (== rows())

  Buffer() {}
  Buffer(int r, int c) : width(c), height(r), rows_(r), cols_(align16(c)), v_((size_t)r * align16(c), T()) {}
  Buffer(int r, int c, T color) : width(c), height(r), rows_(r), cols_(align16(c)), v_((size_t)r * align16(c), color) {}

  // contents unspecified, as the reference's Buffer(r, c) leaves them (buffer.hpp:145: an Eigen array)
  static Buffer uninitialized(int r, int c) {
    Buffer b;
    b.width = c;
    b.height = r;
    b.rows_ = r;
    b.cols_ = align16(c);
    b.v_.resize((size_t)r * align16(c));
    return b;
  }

  int rows() const { return rows_; }
  int cols() const { return cols_; }
  T* data() { return v_.data(); }
  const T* data() const { return v_.data(); }
  size_t size() const { return v_.size(); }

  // Eigen-style resize: exact dimensions, contents unspecified (here: zero)
  void resize(int r, int c) {
    rows_ = r;
    cols_ = c;
    v_.assign((size_t)r * c, T());
  }
  // keeps the top-left block
  void conservativeResize(int r, int c) {
    std::vector<T, detail::default_init_allocator<T>> n((size_t)r * c, T());
    const int rr = std::min(r, rows_), cc = std::min(c, cols_);
    for (int y = 0; y < rr; ++y)
      for (int x = 0; x < cc; ++x) n[(size_t)y * c + x] = v_[(size_t)y * cols_ + x];
    v_.swap(n);
    rows_ = r;
    cols_ = c;
  }

  T& operator()(int row, int col) { return v_[(size_t)row * cols_ + col]; }
  const T& operator()(int row, int col) const { return v_[(size_t)row * cols_ + col]; }
  // linear index in storage (row-major) order, as Eigen's single-index access on the reference's Buffer
  T& operator()(int i) { return v_[(size_t)i]; }
  const T& operator()(int i) const { return v_[(size_t)i]; }
  // buffer.hpp:534-544: note that the patch ROW follows the image X offset (patch(ix, iy) = pixel(x+ix-s/2, y+iy-s/2))
  void getPatch(Buffer<uint8_t>& patch, int x, int y, int size) const {
    patch.resize(size, size);
    for (int ix = 0; ix < size; ix++)
      for (int iy = 0; iy < size; iy++) patch(ix, iy) = getPixel(x + ix - (size / 2), y + iy - (size / 2));
  }
  void setPixel(int x, int y, T color) { v_[(size_t)y * cols_ + x] = color; }
  T getPixel(int x, int y) const { return v_[(size_t)y * cols_ + x]; }
  void set(T color) { std::fill(v_.begin(), v_.end(), color); }
  Dimension getDimension() const { return Dimension(cols_, rows_); }

  // buffer.hpp:630-654
  void clearBoundary() {
    const int h = height, w = width, wa = cols_;
    T* p = data();
    for (int x = 0; x < 2; ++x)
      for (int y = 0; y < h; ++y) p[(size_t)y * wa + x] = T();
    for (int x = 0; x < w; ++x) p[x] = T();
    for (int x = 0; x < w; ++x)
      for (int y = h - 2; y < h; ++y) p[(size_t)y * wa + x] = T();
    for (int y = 0; y < h; ++y) p[(size_t)y * wa + (wa - 1)] = T();
  }

  // buffer.hpp:197-318.  0 = ok, 1 = error (message on stdout).  Gray stays, RGB becomes
  // (r+g+b)/3, columns are padded (with zeros) to the next multiple of 16.
  int readPNG(std::string filename) {
    pngio::Image img;
    const int rc = pngio::decode_file(filename, img);
    if (rc == 1) {
      std::cout << "ERR: File" << filename << " could not be opened for reading" << std::endl;
      return 1;
    }
    if (rc == 2) {
      std::cout << "ERR: File" << filename << " is not recognized as a PNG file" << std::endl;
      return 1;
    }
    if (rc != 0) {
      std::cout << "ERR: Error during read_image" << std::endl;
      return 1;
    }
    width = img.width;
    height = img.height;
    resize(height, width);
    const bool gray = img.color_type == 0, rgb = img.color_type == 2;
    if (img.bit_depth == 16) {
      for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
          const uint8_t* row = &img.pixels[(size_t)y * width * img.channels * 2];
          (*this)(y, x) = (T)(((int)row[x * 2] << 8) + row[x * 2 + 1]);
        }
    } else if (gray || rgb) {
      for (int y = 0; y < height; ++y) {
        const uint8_t* row = &img.pixels[(size_t)y * width * img.channels];
        for (int x = 0; x < width; ++x)
          (*this)(y, x) = gray ? (T)row[x] : (T)((row[3 * x] + row[3 * x + 1] + row[3 * x + 2]) / 3);
      }
    }
    conservativeResize(height, align16(width));
    if (!(gray || rgb)) {
      std::cout << "ERR: found something other than gray or 3 channel color image(" << img.color_type
                << ") aborting!" << std::endl;
      return 1;
    }
    return 0;
  }

  // buffer.hpp:319-393: 8-bit gray, visible region only
  void writePNG(std::string filename) {
    std::vector<uint8_t> px((size_t)width * height);
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) px[(size_t)y * width + x] = (uint8_t)(*this)(y, x);
    if (pngio::encode_file(filename, px.data(), width, height, 1))
      std::cout << "ERR: File" << filename << " could not be opened for writing" << std::endl;
  }

  // buffer.hpp:395-474: 8-bit RGB, visible region only (instantiated for T = RGBColor)
  void writePNGRGB(std::string filename) {
    std::vector<uint8_t> px((size_t)width * height * 3);
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) {
        const T& c = (*this)(y, x);
        uint8_t* o = &px[((size_t)y * width + x) * 3];
        o[0] = c.r;
        o[1] = c.g;
        o[2] = c.b;
      }
    if (pngio::encode_file(filename, px.data(), width, height, 3))
      std::cout << "ERR: File" << filename << " could not be opened for writing" << std::endl;
  }

  Buffer<RGBColor> convertToRGB() const;

 private:
  int rows_ = 0, cols_ = 0;
  std::vector<T, detail::default_init_allocator<T>> v_;
};

template <class T>
inline Buffer<RGBColor> Buffer<T>::convertToRGB() const {
  Buffer<RGBColor> out(rows_, cols_);
  out.width = width;
  for (int y = 0; y < rows_; ++y)
    for (int x = 0; x < cols_; ++x) {
      const uint8_t c = (uint8_t)(*this)(y, x);
      out(y, x) = RGBColor(c, c, c);
    }
  return out;
}

// Piecewise-linear KITTI colour ramp as used by getDisparityVisualization (buffer.hpp:958-1010):
// value in [0, 0.8] -> RGB.  Kept in the same float arithmetic so the bytes are identical.
inline RGBColor disparityColor(float disp) {
  static const float ramp[8][4] = {{0, 0, 1, 185}, {1, 0, 0, 114}, {1, 0, 1, 174}, {0, 1, 0, 114},
                                   {0, 1, 1, 185}, {1, 1, 0, 114}, {1, 1, 1, 0},   {0, 0, 0, 114}};
  float sum = 0;
  for (int i = 0; i < 8; ++i) sum += ramp[i][3];
  float weights[8], cumsum[8];
  cumsum[0] = 0;
  for (int i = 0; i < 7; ++i) {
    weights[i] = sum / ramp[i][3];
    cumsum[i + 1] = cumsum[i] + ramp[i][3] / sum;
  }
  const float value = std::max(0.f, std::min(0.8f, (disp - 0.f) / (128.f - 0.f)));
  int bin;
  for (bin = 0; bin < 7; ++bin)
    if (value < cumsum[bin + 1]) break;
  const float w = 1.0f - (value - cumsum[bin]) * weights[bin];
  return RGBColor(static_cast<uint8_t>((w * ramp[bin][0] + (1.0f - w) * ramp[bin + 1][0]) * 255.0f),
                  static_cast<uint8_t>((w * ramp[bin][1] + (1.0f - w) * ramp[bin + 1][1]) * 255.0f),
                  static_cast<uint8_t>((w * ramp[bin][2] + (1.0f - w) * ramp[bin + 1][2]) * 255.0f));
}

// buffer.hpp:949-1014: gray image as RGB with every support painted in its disparity colour
inline Buffer<RGBColor> getDisparityVisualization(Buffer<uint8_t>& srcImg, std::vector<Support>& support) {
  Buffer<RGBColor> vis = srcImg.convertToRGB();
  for (auto& s : support) vis.setPixel(s.x, s.y, disparityColor(s.d));
  return vis;
}

}  // namespace ndb
#endif
