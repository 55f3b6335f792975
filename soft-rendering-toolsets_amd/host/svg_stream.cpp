// See svg_stream.h. Host element walk for the MI355X rasterizer path.
//
// Every arithmetic step that decides WHERE a primitive lands (transform stack, float
// narrowing at the rasterize_* call boundaries) follows the reference expression by
// expression, because the ordered stream must be identical to the sequence of
// rasterize_triangle / rasterize_line / rasterize_point / rasterize_image calls
// SoftwareRendererImp::draw_svg makes (Assignments/DrawSVG/src/software_renderer.cpp:17-52, 94-265).
#include "svg_stream.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <utility>

#include "triangulation.h"  // the reference's ear-clipper, reused as-is

namespace CMU462 {

// SVGRenderer::transform (svg_renderer.h:37-48) for the current `transformation`, its nine entries held in locals:
//   u = M * (p.x, p.y, 1) = p.x * col0 + p.y * col1 + 1.0 * col2   (matrix3x3.cpp:138-142, summed left to right; 1.0 * c == c)
//   return (u.x / u.z, u.y / u.z)
// The same double operations in the same order - the stream is byte-identical to the one the member function gives
// (set_reference_transforms(true) selects that one; tests/test_dropin_cpu.py compares the two on every SVG of the reference) -
// without the out-of-line Matrix3x3::operator* and its Vector3D temporaries: 20 ns -> 3 ns per point, and test3.svg has 9870.
struct SvgStreamBuilder::PointMap {
  double c0x, c0y, c0z, c1x, c1y, c1z, c2x, c2y, c2z;
  explicit PointMap(const Matrix3x3& m)
      : c0x(m[0].x), c0y(m[0].y), c0z(m[0].z), c1x(m[1].x), c1y(m[1].y), c1z(m[1].z), c2x(m[2].x), c2y(m[2].y), c2z(m[2].z) {}
  inline Vector2D operator()(const Vector2D& p) const {
    const double ux = (p.x * c0x + p.y * c1x) + c2x;
    const double uy = (p.x * c0y + p.y * c1y) + c2y;
    const double uz = (p.x * c0z + p.y * c1z) + c2z;
    return Vector2D(ux / uz, uy / uz);
  }
};

const std::vector<srt_prim>& SvgStreamBuilder::build(SVG& svg, const Matrix3x3& svg_2_screen, size_t sample_rate) {
  stream_.clear();
  if (stream_.capacity() < last_size_ + 16) stream_.reserve(last_size_ + 16);
  textures_.clear();
  sample_rate_ = sample_rate;
  transformation = svg_2_screen;

  for (size_t i = 0; i < svg.elements.size(); ++i) walk(svg.elements[i]);

  // Canvas outline: four corners pushed one pixel outwards, then four lines (cpp:31-48).
  Vector2D a = transform(Vector2D(0, 0));
  Vector2D b = transform(Vector2D(svg.width, 0));
  Vector2D c = transform(Vector2D(0, svg.height));
  Vector2D d = transform(Vector2D(svg.width, svg.height));
  a.x--; a.y--;
  b.x++; b.y--;
  c.x--; c.y++;
  d.x++; d.y++;
  emit_line(a.x, a.y, b.x, b.y, Color::Black);
  emit_line(a.x, a.y, c.x, c.y, Color::Black);
  emit_line(d.x, d.y, b.x, b.y, Color::Black);
  emit_line(d.x, d.y, c.x, c.y, Color::Black);
  last_size_ = stream_.size();
  return stream_;
}

void SvgStreamBuilder::walk(SVGElement* e) {
  // Push: T <- T * M_e. Pop: T <- T * inv(M_e); the pop is NOT an exact undo in floating
  // point and later siblings see the drifted matrix, exactly as in the reference (cpp:102,133).
  transformation = transformation * e->transform;

  switch (e->type) {
    case POINT: {
      Point& pt = static_cast<Point&>(*e);
      Vector2D p = transform(pt.position);
      emit_point(p.x, p.y, pt.style.fillColor);
    } break;

    case LINE: {
      Line& ln = static_cast<Line&>(*e);
      Vector2D p0 = transform(ln.from);
      Vector2D p1 = transform(ln.to);
      emit_line(p0.x, p0.y, p1.x, p1.y, ln.style.strokeColor);
    } break;

    case POLYLINE: {
      Polyline& pl = static_cast<Polyline&>(*e);
      Color c = pl.style.strokeColor;
      if (c.a != 0) {
        // (every point is the end of one segment and the start of the next: transformed once - transform() is a pure function
        //  of the point and the matrix, so the shared value is the value the reference computes twice)
        const std::vector<Vector2D>& q = transformed(pl.points);
        int n = (int)pl.points.size();
        for (int i = 0; i + 1 < n; i++) emit_line(q[i].x, q[i].y, q[i + 1].x, q[i + 1].y, c);
      }
    } break;

    case RECT: {
      Rect& rc = static_cast<Rect&>(*e);
      // corner coordinates pass through float first (cpp:176-184)
      float x = rc.position.x, y = rc.position.y;
      float w = rc.dimension.x, h = rc.dimension.y;
      Vector2D p0 = transform(Vector2D(x, y));
      Vector2D p1 = transform(Vector2D(x + w, y));
      Vector2D p2 = transform(Vector2D(x, y + h));
      Vector2D p3 = transform(Vector2D(x + w, y + h));
      Color c = rc.style.fillColor;
      if (c.a != 0) {
        emit_triangle(p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, c);
        emit_triangle(p2.x, p2.y, p1.x, p1.y, p3.x, p3.y, c);
      }
      c = rc.style.strokeColor;
      if (c.a != 0) {
        emit_line(p0.x, p0.y, p1.x, p1.y, c);
        emit_line(p1.x, p1.y, p3.x, p3.y, c);
        emit_line(p3.x, p3.y, p2.x, p2.y, c);
        emit_line(p2.x, p2.y, p0.x, p0.y, c);
      }
    } break;

    case POLYGON: {
      Polygon& pg = static_cast<Polygon&>(*e);
      Color c = pg.style.fillColor;
      if (c.a != 0) {
        // triangulate() depends on the polygon's own points only, not on the view: a redraw after a pan or zoom reuses the list
        // (the reference ear-clips again every frame; for BASELINE configs[1] that is 2.8 ms of a 3.2 ms redraw).  The cache entry
        // is checked against the points themselves, so an edited polygon is triangulated again.
        const CachedTriangulation& t = triangulation_of(pg);
        if (!t.index.empty() && !reference_transforms_) {
          // the ear-clipper hands back copies of the polygon's own points: transform each point once and address the corners
          // by index (a 1000-gon's 998 triangles name 2994 corners, and the outline names every point twice more)
          const std::vector<Vector2D>& q = transformed(pg.points);
          for (size_t i = 0; i + 2 < t.index.size(); i += 3) {
            const Vector2D &p0 = q[t.index[i]], &p1 = q[t.index[i + 1]], &p2 = q[t.index[i + 2]];
            emit_triangle(p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, c);
          }
        } else {
          const std::vector<Vector2D>& tris = t.tris;
          for (size_t i = 0; i + 2 < tris.size(); i += 3) {
            Vector2D p0 = transform(tris[i]);
            Vector2D p1 = transform(tris[i + 1]);
            Vector2D p2 = transform(tris[i + 2]);
            emit_triangle(p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, c);
          }
        }
      }
      c = pg.style.strokeColor;
      if (c.a != 0) {
        const std::vector<Vector2D>& q = transformed(pg.points);
        int n = (int)pg.points.size();
        for (int i = 0; i < n; i++) {
          const Vector2D &p0 = q[i], &p1 = q[(i + 1) % n];
          emit_line(p0.x, p0.y, p1.x, p1.y, c);
        }
      }
    } break;

    case GROUP: {
      Group& g = static_cast<Group&>(*e);
      for (size_t i = 0; i < g.elements.size(); ++i) walk(g.elements[i]);
    } break;

    case IMAGE: {  // draw_image (cpp:249-256): the double corners narrow to rasterize_image's float parameters
      Image& im = static_cast<Image&>(*e);
      Vector2D p0 = transform(im.position);
      Vector2D p1 = transform(im.position + im.dimension);
      emit_image(p0.x, p0.y, p1.x, p1.y, im.tex);
    } break;

    case ELLIPSE:  // draw_ellipse is empty in the reference (cpp:243-247)
    default:
      break;
  }

  transformation = transformation * e->transform.inv();
}

// The points of one element under the current matrix (scratch, valid until the next call).
const std::vector<Vector2D>& SvgStreamBuilder::transformed(const std::vector<Vector2D>& points) {
  scratch_.resize(points.size());
  if (reference_transforms_) {
    for (size_t i = 0; i < points.size(); i++) scratch_[i] = transform(points[i]);
  } else {
    const PointMap map(transformation);
    for (size_t i = 0; i < points.size(); i++) scratch_[i] = map(points[i]);
  }
  return scratch_;
}

const SvgStreamBuilder::CachedTriangulation& SvgStreamBuilder::triangulation_of(const Polygon& pg) {
  CachedTriangulation& c = tri_cache_[&pg];
  bool same = c.valid && c.points.size() == pg.points.size();
  for (size_t i = 0; same && i < pg.points.size(); i++) same = c.points[i].x == pg.points[i].x && c.points[i].y == pg.points[i].y;
  if (!same) {
    c.points = pg.points;
    c.tris.clear();
    triangulate(pg, c.tris);
    // which point each corner is: the ear-clipper copies corners out of pg.points (triangulation.cpp), so every corner has a
    // point with the very same two doubles (any of several equal points will do: equal points transform equally).  A corner
    // without one - NaN coordinates never compare equal - leaves the index empty and the corners are transformed one by one.
    c.index.clear();
    c.index.reserve(c.tris.size());
    size_t hint = 0;
    const size_t n = pg.points.size();
    for (size_t i = 0; i < c.tris.size(); i++) {
      size_t k = 0;
      for (; k < n; k++) {
        const size_t j = hint + k < n ? hint + k : hint + k - n;
        if (pg.points[j].x == c.tris[i].x && pg.points[j].y == c.tris[i].y) { hint = j; break; }
      }
      if (k == n) { c.index.clear(); break; }
      c.index.push_back((uint32_t)hint);
    }
    c.valid = true;
  }
  return c;
}

void SvgStreamBuilder::emit_triangle(float x0, float y0, float x1, float y1, float x2, float y2,
                                        const Color& c) {
  stream_.emplace_back();                 // (value-initialised: every byte of the record zero, then filled in place)
  srt_prim& p = stream_.back();
  p.kind = SRT_PRIM_TRIANGLE;
  p.v.tri[0] = x0; p.v.tri[1] = y0;
  p.v.tri[2] = x1; p.v.tri[3] = y1;
  p.v.tri[4] = x2; p.v.tri[5] = y2;
  p.rgba[0] = c.r; p.rgba[1] = c.g; p.rgba[2] = c.b; p.rgba[3] = c.a;
}

void SvgStreamBuilder::emit_point(double x, double y, const Color& c) {
  stream_.emplace_back();
  srt_prim& p = stream_.back();
  p.kind = SRT_PRIM_POINT;
  p.v.point[0] = x;
  p.v.point[1] = y;
  p.rgba[0] = c.r; p.rgba[1] = c.g; p.rgba[2] = c.b; p.rgba[3] = c.a;
}

void SvgStreamBuilder::emit_image(float x0, float y0, float x1, float y1, const Texture& tex) {
  stream_.emplace_back();
  srt_prim& p = stream_.back();
  p.kind = SRT_PRIM_IMAGE;
  size_t id = 0;
  while (id < textures_.size() && textures_[id] != &tex) id++;
  if (id == textures_.size()) textures_.push_back(&tex);
  p.reserved = (uint32_t)id;
  p.v.tri[0] = x0; p.v.tri[1] = y0; p.v.tri[2] = x1; p.v.tri[3] = y1;
}

// rasterize_line(x0, y0, x1, y1, color) (cpp:303-318): ONE record; the device expands rasterize_line_xiaolinwu into its
// rasterize_point calls (csrc/raster.hip: raster_setup + the tile kernel).  The double coordinates narrow to the call's float
// parameters here, as at the reference's call sites; the stroke alpha travels along but is replaced by the Wu coverage.
void SvgStreamBuilder::emit_line(float x0, float y0, float x1, float y1, Color color) {
  stream_.emplace_back();
  srt_prim& p = stream_.back();
  p.kind = SRT_PRIM_LINE;
  p.v.tri[0] = x0; p.v.tri[1] = y0; p.v.tri[2] = x1; p.v.tri[3] = y1;
  p.rgba[0] = color.r; p.rgba[1] = color.g; p.rgba[2] = color.b; p.rgba[3] = color.a;
}

}  // namespace CMU462
