"""Pin the constants this build hard-codes against the reference's source TEXT (no reference code is run).
Only in the build container: skipped when /root/reference is absent (the GPU box)."""
import os
import re

import pytest

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def text(name):
    return open(os.path.join(REF, name), errors="ignore").read()


def test_uniform_defaults_match_the_c_abi():
    import shaderbox_amd
    from shaderbox_amd import build
    build.build(verbose=False)
    ub = text("uniform_buffer.h")
    got = dict((m.group(2), m.group(3)) for m in re.finditer(r"_uniform\((\w+),\s*(\w+),\s*([^)]*\)?)\)", ub))
    a = shaderbox_amd.clouds_defaults()
    assert got["wind_dir"].replace(" ", "") == "vec3(0,0,.2)" and list(a.wind_dir)[2] == pytest.approx(.2)
    assert got["sun_dir"].replace(" ", "") == "vec3(0,0,-1)" and list(a.sun_dir) == [0.0, 0.0, -1.0]
    assert got["sun_color"].replace(" ", "") == "vec3(1.,.7,.55)"
    for name, val in [("sun_power", 8.0), ("cld_march_steps", 100), ("illum_march_steps", 6), ("sigma_scattering", .15),
                      ("cld_coverage", .535), ("cld_thick", 125.0), ("atm_radius", 5000.0), ("atm_ground_y", 4750.0)]:
        assert float(got[name].strip("()")) == pytest.approx(val)
        assert float(getattr(a, name)) == pytest.approx(val)
    b = shaderbox_amd.sdf_ao_defaults()
    assert float(got["fog_density"].strip("()")) == pytest.approx(b.fog_density)
    assert float(got["fog_falloff"].strip("()")) == pytest.approx(b.fog_falloff)


@pytest.mark.parametrize("fname,needles", [
    ("noise_iq.h", ["753.5453123", "p.y*157.0", "113.0*p.z", "hash(n+270.0)", "hash(n+271.0)"]),
    ("noise_worley.h", ["43758.5453123", "127.1, 311.7, 74.7", "269.5, 183.3, 246.1", "113.5, 271.9, 124.6", "1.0, 57.0, 113.0"]),
    ("app_clouds.h", ["#define hg_g (.2)", "#define cld_noise_factor .001", "pos * 2.03, 2.64, .5, .5", "cov + .0135",
                      "density < .005", "projection * 150.", "cloud.alpha > .999", "< 0.05) return sky", "1500.0) * 5.0", "10.0) * .6",
                      "DECL_FBM_FUNC(fbm, 4, noise_iq(p))", "#define FOV 1."]),
    ("app_egg.h", ["const int steps = 80;", "const float end = 15.;", "const int steps = 20;", "#define EPSILON 0.001",
                   "u_time * -100.0", "pedal_speed = 400.", "eye = vec3(.0, .25, 5.25)", "#define BAR_SEPARATION 0.6"]),
    ("app_raytracer.h", ["for (int i = 0; i < 2; i++)", "color *= 0.1", "reflect(hit.normal, ray.direction)", "tan(radians(30.))",
                         "2.333 * cb_plane_dist"]),
    ("cornell_box.h", ["#define cb_plane_dist 2.", "vec3(0.7913, 0.7913, 0.7913)", "vec3(0.6795, 0.0612, 0.0529)",
                       "vec3(0.1878, 0.1274, 0.4287)", "ior = 1.333"]),
    ("app_atmosphere.h", ["#define hg_g (.76)", "5.5e-6, 13.0e-6, 22.4e-6", "hR = 7994.0", "hM = 1200.0", "earth_radius = 6360e3",
                          "atmosphere_radius = 6420e3", "num_samples = 16", "num_samples_light = 8", "sun_power = 20.0",
                          "#define FROM_SPACE 1", "betaM * 1.1"]),
    ("app_sdf_ao.h", ["const int steps = 70;", "const float end = 20.;", "d.x < .005", "size = vec3(1.3, 1., 1.25)",
                      "rotate_around_x(-90.)", "rotate_around_y(180.)", "u_time * 50."]),
    ("app_planet.h", ["#define TERR_STEPS 120", "#define TERR_EPS .005", "const int steps = 75;", "#define vol_coeff_absorb 30.034",
                      "#define cld_coverage .29475675", "#define cld_fuzzy .0335", "df.x * .4567", "pos * 2.0987, 2.0244, .454, .454",
                      "cloud.pos * 3.2343 + vec3(.35, 13.35, 2.67)", "rotate_around_y(27.)", "u_time * -12.", "u_time * 8."]),
    ("app_vinyl.h", ["\t\t60;", "const float end = 40.;", "u_time * 200.", "u_time * 3.6758", ".07575", "rr * 24.", "hit.origin * 2.456"]),
    ("volumetric.h", ["(4. + PI) * pow(1. + hg_g*hg_g - 2.*hg_g*mu, 1.5)", "3. * (1. + mu*mu)", "(16. * PI)"]),
    ("def.h", ["#define PI 3.14159265359", "#define BIAS 1e-4", "#define max_dist 1e8", "float(max_dist + 1e1)"]),
    ("util.h", ["const float p = 1. / 2.2;", "mod(pattern.x + pattern.y, 2.0)"]),
    ("main.h", ["fragCoord.xy / u_res.xy", "(2.0 * point_ndc - 1.0) * aspect_ratio * FOV", "vec4(linear_to_srgb(color), 1)"]),
])
def test_literals_this_build_relies_on_are_in_the_reference_text(fname, needles):
    src = text(fname)
    for n in needles:
        assert n in src, (fname, n)
