"""Host mirror of the reference's Parser (Parser.py:15-71): same CLI, same scene
text format, same index assignment (one counter shared by sph/pln/box AND lgt
lines, starting at 1; material ids from 1; surfaces[0] is the Background)."""
import argparse

from Camera import Camera
from Light import Light
from Material import Material
from SceneSettings import SceneSettings
from surfaces.Background import Background
from surfaces.Cube import Cube
from surfaces.InfinitePlane import InfinitePlane
from surfaces.Sphere import Sphere


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Python Ray Tracer (MI355X-native hot path)')
    parser.add_argument('scene_file', type=str, help='Path to the scene file')
    parser.add_argument('output_image', type=str, help='Name of the output image file')
    parser.add_argument('--width', type=int, default=500, help='Image width')
    parser.add_argument('--height', type=int, default=500, help='Image height')
    # optional extras (absent in the reference; defaults keep its behaviour)
    parser.add_argument('--seed', type=int, default=0, help='seed of the soft-shadow jitter hash')
    parser.add_argument('--no-jitter', action='store_true', help='deterministic shadows (jitter forced to 0)')
    parser.add_argument('--shadow-rays', type=int, default=None, help='override root_number_shadow_rays')
    parser.add_argument('--frames', type=int, default=1,
                        help='average this many stochastic frames (seeds seed, seed+1, ...) on the GPU')
    parser.add_argument('--python-ingest', action='store_true',
                        help='build the scene through Python objects and the recursive host tree build '
                             '(default: C++ tokenizer + GPU tree build, identical tables)')
    return parser.parse_args(argv)


def parse_scene_file(file_path):
    """-> (camera, scene_settings, objects_3D (always []), surfaces, materials, lights)."""
    camera = None
    scene_settings = None
    surfaces = [Background()]
    materials = []
    lights = []
    counter = {"index": 1, "mat": 1}

    def next_index():
        k = counter["index"]
        counter["index"] = k + 1
        return k

    def on_cam(v):
        nonlocal camera
        camera = Camera(v[:3], v[3:6], v[6:9], v[9], v[10])

    def on_set(v):
        nonlocal scene_settings
        scene_settings = SceneSettings(v[:3], v[3], v[4])
        # the background material is created here, at list index 0 after the sort (Parser.py:46-47)
        materials.append(Material(scene_settings.background_color, [0, 0, 0], [0, 0, 0], 0, 0, 0))

    def on_mtl(v):
        materials.append(Material(v[:3], v[3:6], v[6:9], v[9], v[10], counter["mat"]))
        counter["mat"] += 1

    handlers = {
        "cam": on_cam,
        "set": on_set,
        "mtl": on_mtl,
        "sph": lambda v: surfaces.append(Sphere(v[:3], v[3], int(v[4]), next_index())),
        "pln": lambda v: surfaces.append(InfinitePlane(v[:3], v[3], int(v[4]), next_index())),
        "box": lambda v: surfaces.append(Cube(v[:3], v[3], int(v[4]), next_index())),
        "lgt": lambda v: lights.append(Light(v[:3], v[3:6], v[6], v[7], v[8], next_index())),
    }
    with open(file_path, 'r') as f:
        for raw in f:
            text = raw.strip()
            if not text or text.startswith("#"):
                continue
            kind, *fields = text.split()
            if kind not in handlers:
                raise ValueError("Unknown object type: {}".format(kind))
            handlers[kind]([float(x) for x in fields])
    materials.sort(key=lambda m: m.index)
    return camera, scene_settings, [], surfaces, materials, lights
