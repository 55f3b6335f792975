// See pathtracer_hip.h.  Host orchestration only: scene flattening through the C ABI, the reference's epoch
// scheme, running-mean accumulation, progress / cancel.  Everything below trace_pixel is on the GPU.
#include "pathtracer_hip.h"

#include <cstring>
#include <limits>

#include "camera_iview.h"

#include "../gui/render.h"

namespace PT {

namespace {

void check(int status, const char* what) {
    if(status != SRT_OK) die("%s failed (%d): %s", what, status, srt_last_error());
}

void mat_to_array(const Mat4& m, float out[16]) {
    for(int i = 0; i < 16; i++) out[i] = m.data[i];
}

srt_pt_material make_material(uint32_t type, Spectrum a, Spectrum b, float ior) {
    srt_pt_material m;
    m.type = type;
    m.a[0] = a.r; m.a[1] = a.g; m.a[2] = a.b;
    m.b[0] = b.r; m.b[1] = b.g; m.b[2] = b.b;
    m.ior = ior;
    return m;
}

} // namespace

static void fatal(const char* what, int status, const char* message) {
    die("%s failed (%d): %s", what, status, message);
}

// Pathtracer::log_ray (rays/pathtracer.cpp:191-193): the rays sample_direct_lighting's 0.0005 coin selected
// (student/pathtracer.cpp:148: log_ray(world_ray_task6, 5.0f), color = Spectrum{1.0f}), recorded by the kernels and replayed
// into the GUI's ray log after every launch, on the render thread (Widget_Render::log_ray takes its own mutex, gui/widgets.cpp:625-628).
void Pathtracer::deliver_logged_rays(void* self, const srt_pt_logged_ray* rays, size_t n) {
    Pathtracer* pt = static_cast<Pathtracer*>(self);
    for(size_t i = 0; i < n; i++) {
        Ray ray;                                           // (the direction is already the unit vector Ray's constructor makes)
        ray.point = Vec3(rays[i].point[0], rays[i].point[1], rays[i].point[2]);
        ray.dir = Vec3(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
        ray.dist_bounds = Vec2(EPS_F, std::numeric_limits<float>::max());
        pt->gui.log_ray(ray, rays[i].t, Spectrum(1.0f));
    }
}

Pathtracer::Pathtracer(Gui::Widget_Render& gui, Vec2) : gui(gui), core(nullptr, 0, fatal) {
    core.set_ray_log(&Pathtracer::deliver_logged_rays, this);
}

Pathtracer::~Pathtracer() {
    core.cancel();
}

void Pathtracer::set_samples(size_t samples) {
    core.set_samples(samples);
}

void Pathtracer::set_params(size_t w, size_t h, size_t samples, size_t depth, bool use_bvh) {
    scene_use_bvh = use_bvh;
    accumulator.resize(w, h);
    core.set_params(w, h, samples, depth);
}

void Pathtracer::build_scene(Scene& layout_scene) {
    for(int r = 0; r < core.ranks(); r++) feed_scene(core.context(r), layout_scene);   // replicated: a few MB per device
    // the BVH<Object> boxes for visualize_bvh: fixed until the next build_scene
    bvh_boxes.clear(); bvh_links.clear();
    float none_f[6]; uint32_t none_u[4];
    const long n = srt_pt_dump_bvh(core.context(0), -1, none_f, none_u, 0, nullptr);
    if(n > 0) {
        bvh_boxes.resize(6 * (size_t)n); bvh_links.resize(4 * (size_t)n);
        srt_pt_dump_bvh(core.context(0), -1, bvh_boxes.data(), bvh_links.data(), (size_t)n, nullptr);
    }
}

// The object walk of the reference's build_scene, feeding the C ABI instead of PT::Object constructors.
void Pathtracer::feed_scene(srt_pt* ctx, Scene& layout_scene) {
    check(srt_pt_scene_begin(ctx), "srt_pt_scene_begin");
    bool warned = false;
    layout_scene.for_items([&, this](Scene_Item& item) {
        if(item.is<Scene_Object>()) {
            Scene_Object& obj = item.get<Scene_Object>();
            const Material::Options& opt = obj.material.opt;
            if(!obj.opt.render) return;

            srt_pt_material mat;
            bool is_light = false;
            switch(opt.type) {
            case Material_Type::lambertian: mat = make_material(SRT_MAT_LAMBERTIAN, opt.albedo.to_linear(), {}, 1.0f); break;
            case Material_Type::mirror: mat = make_material(SRT_MAT_MIRROR, opt.reflectance, {}, 1.0f); break;
            case Material_Type::refract: mat = make_material(SRT_MAT_REFRACT, opt.transmittance, {}, opt.ior); break;
            case Material_Type::glass: mat = make_material(SRT_MAT_GLASS, opt.transmittance, opt.reflectance, opt.ior); break;
            case Material_Type::diffuse_light:
                mat = make_material(SRT_MAT_DIFFUSE_LIGHT, obj.material.emissive(), {}, 1.0f);
                is_light = true;
                break;
            default: return;
            }
            uint32_t idx = 0;
            check(srt_pt_add_material(ctx, &mat, &idx), "srt_pt_add_material");

            float T[16];
            mat_to_array(obj.pose.transform(), T);
            auto add_mesh = [&](const GL::Mesh& mesh, bool light) {
                std::vector<float> pos, nrm;
                for(const auto& v : mesh.verts()) {
                    pos.insert(pos.end(), {v.pos.x, v.pos.y, v.pos.z});
                    nrm.insert(nrm.end(), {v.norm.x, v.norm.y, v.norm.z});
                }
                const auto& idxs = mesh.indices();
                check(srt_pt_add_mesh(ctx, pos.data(), nrm.data(), (uint32_t)mesh.verts().size(), idxs.data(),
                                      (uint32_t)idxs.size(), T, idx, light ? 1 : 0),
                      "srt_pt_add_mesh");
            };
            if(obj.is_shape()) {
                const float radius = obj.opt.shape.get<PT::Sphere>().radius;
                if(is_light) {
                    // The reference lights a shape through its triangle approximation - area_lights gets
                    // Tri_Mesh(obj.opt.shape.mesh(), false), rays/pathtracer.cpp:110-111 - but intersects the analytic shape.
                    const GL::Mesh mesh = obj.opt.shape.mesh();
                    std::vector<float> pos, nrm;
                    for(const auto& v : mesh.verts()) {
                        pos.insert(pos.end(), {v.pos.x, v.pos.y, v.pos.z});
                        nrm.insert(nrm.end(), {v.norm.x, v.norm.y, v.norm.z});
                    }
                    const auto& idxs = mesh.indices();
                    check(srt_pt_add_sphere_light(ctx, radius, T, idx, pos.data(), nrm.data(), (uint32_t)mesh.verts().size(),
                                                  idxs.data(), (uint32_t)idxs.size()),
                          "srt_pt_add_sphere_light");
                } else {
                    check(srt_pt_add_sphere(ctx, radius, T, idx), "srt_pt_add_sphere");
                }
            } else {
                add_mesh(obj.posed_mesh(), is_light);
            }
        } else if(item.is<Scene_Light>()) {
            // build_lights (rays/pathtracer.cpp:26-64): directional / point / spot lights become delta lights
            const Scene_Light& light = item.get<Scene_Light>();
            const Spectrum r = light.radiance();
            const float rad[3] = {r.r, r.g, r.b};
            const float ab[2] = {light.opt.angle_bounds.x, light.opt.angle_bounds.y};
            const Mat4 pose = light.pose.transform();
            switch(light.opt.type) {
            case Light_Type::directional:
                check(srt_pt_add_light(ctx, SRT_LIGHT_DIRECTIONAL, rad, ab, pose.data), "srt_pt_add_light");
                break;
            case Light_Type::point:
                check(srt_pt_add_light(ctx, SRT_LIGHT_POINT, rad, ab, pose.data), "srt_pt_add_light");
                break;
            case Light_Type::spot:
                check(srt_pt_add_light(ctx, SRT_LIGHT_SPOT, rad, ab, pose.data), "srt_pt_add_light");
                break;
            case Light_Type::sphere:
                if(light.opt.has_emissive_map) {   // Env_Map(light.emissive_copy())
                    const HDR_Image img = light.emissive_copy();
                    const auto [iw, ih] = img.dimension();
                    std::vector<float> rgb(3 * iw * ih);
                    for(size_t i = 0; i < iw * ih; i++) {
                        const Spectrum px = img.at(i);
                        rgb[3 * i] = px.r; rgb[3 * i + 1] = px.g; rgb[3 * i + 2] = px.b;
                    }
                    check(srt_pt_set_env_map(ctx, (uint32_t)iw, (uint32_t)ih, rgb.data()), "srt_pt_set_env_map");
                } else {
                    check(srt_pt_set_env_light(ctx, SRT_ENV_SPHERE, rad), "srt_pt_set_env_light");
                }
                break;
            case Light_Type::hemisphere:
                check(srt_pt_set_env_light(ctx, SRT_ENV_HEMISPHERE, rad), "srt_pt_set_env_light");
                break;
            default:
                break;
            }
        } else if(item.is<Scene_Particles>()) {
            // build_scene, rays/pathtracer.cpp:134-156: one Lambertian copy of the particle mesh per particle,
            // posed by translate(p.pos) * scale(opt.scale)
            Scene_Particles& particles = item.get<Scene_Particles>();
            srt_pt_material mat;
            std::memset(&mat, 0, sizeof mat);
            mat.type = SRT_MAT_LAMBERTIAN;
            const Spectrum albedo = particles.opt.color.to_linear();
            mat.a[0] = albedo.r; mat.a[1] = albedo.g; mat.a[2] = albedo.b;
            uint32_t idx = 0;
            check(srt_pt_add_material(ctx, &mat, &idx), "srt_pt_add_material");
            const GL::Mesh& mesh = particles.mesh();
            std::vector<float> pos, nrm;
            for(const auto& v : mesh.verts()) {
                pos.insert(pos.end(), {v.pos.x, v.pos.y, v.pos.z});
                nrm.insert(nrm.end(), {v.norm.x, v.norm.y, v.norm.z});
            }
            const auto& idxs = mesh.indices();
            for(const Scene_Particles::Particle& p : particles.get_particles()) {
                float T[16];
                mat_to_array(Mat4::translate(p.pos) * Mat4::scale(Vec3{particles.opt.scale}), T);
                check(srt_pt_add_mesh(ctx, pos.data(), nrm.data(), (uint32_t)mesh.verts().size(), idxs.data(),
                                      (uint32_t)idxs.size(), T, idx, 0),
                      "srt_pt_add_mesh");
            }
        }
    });
    check(srt_pt_scene_commit(ctx, scene_use_bvh ? 1 : 0), "srt_pt_scene_commit");
}

void Pathtracer::begin_render(Scene& layout_scene, const Camera& cam, bool add_samples) {
    core.cancel();
    if(!add_samples) {
        const auto t0 = std::chrono::steady_clock::now();
        build_scene(layout_scene);
        core.note_build_time(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
    }
    // the matrix Camera::generate_ray itself uses (camera_iview.h): bit-identical camera rays, no header patch
    float iview[16];
    mat_to_array(srt_host::camera_iview(cam), iview);
    core.begin(iview, cam.get_fov(), cam.get_ar(), add_samples);
}

void Pathtracer::cancel() {
    core.cancel();
}

bool Pathtracer::in_progress() const {
    return core.in_progress();
}

float Pathtracer::progress() const {
    return core.progress();
}

std::pair<float, float> Pathtracer::completion_time() const {
    return core.completion_time();
}

const HDR_Image& Pathtracer::get_output() {
    core.copy_accumulator(accumulator_copy);
    auto [w, h] = accumulator.dimension();
    for(size_t i = 0; i < w * h && 3 * i + 2 < accumulator_copy.size(); i++)
        accumulator.at(i) = Spectrum(accumulator_copy[3 * i], accumulator_copy[3 * i + 1], accumulator_copy[3 * i + 2]);
    return accumulator;
}

void Pathtracer::tonemap_to(std::vector<unsigned char>& data, float exposure) {
    core.tonemap(data, exposure);
}

const GL::Tex2D& Pathtracer::get_output_texture(float exposure) {
    tonemap_to(tonemap_out, exposure);
    output_tex.image((int)core.width(), (int)core.height(), tonemap_out.data());
    return output_tex;
}

// Boxes of the BVH<Object> built for the GPU (same node arrays as the reference's, student/bvh.inl:324-372).
size_t Pathtracer::visualize_bvh(GL::Lines& lines, GL::Lines& active, size_t level) {
    const std::vector<float>& boxes = bvh_boxes;        // every node of the tree (build_scene sized them by the node count)
    const std::vector<uint32_t>& links = bvh_links;
    const size_t n = links.size() / 4;
    if(n == 0) return 0;
    size_t max_level = 0;
    std::vector<std::pair<uint32_t, size_t>> stack{{0u, size_t(0)}};
    while(!stack.empty()) {
        auto [idx, lvl] = stack.back();
        stack.pop_back();
        max_level = std::max(max_level, lvl);
        const float* b = &boxes[6 * idx];
        Vec3 mn(b[0], b[1], b[2]), mx(b[3], b[4], b[5]);
        Vec3 color = lvl == level ? Vec3(1.0f, 0.0f, 0.0f) : Vec3(1.0f);
        GL::Lines& add = lvl == level ? active : lines;
        const Vec3 c[8] = {Vec3(mn.x, mn.y, mn.z), Vec3(mx.x, mn.y, mn.z), Vec3(mn.x, mx.y, mn.z), Vec3(mn.x, mn.y, mx.z),
                           Vec3(mx.x, mx.y, mn.z), Vec3(mn.x, mx.y, mx.z), Vec3(mx.x, mn.y, mx.z), Vec3(mx.x, mx.y, mx.z)};
        const int e[12][2] = {{0, 1}, {0, 2}, {0, 3}, {7, 5}, {7, 6}, {7, 4}, {2, 4}, {2, 5}, {3, 6}, {3, 5}, {1, 4}, {1, 6}};
        for(auto& ed : e) add.add(c[ed[0]], c[ed[1]], color);
        const uint32_t l = links[4 * idx + 2], r = links[4 * idx + 3];
        if(l != r && l < n && r < n) {
            stack.push_back({l, lvl + 1});
            stack.push_back({r, lvl + 1});
        }
    }
    return max_level;
}

} // namespace PT
