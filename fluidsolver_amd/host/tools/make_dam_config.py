#!/usr/bin/env python3
"""Writes the dam-break scenes of fluidsolver_amd.scenes.dam_break as <MultiScene> blocks plus their
tank files, so that the C++ host layer and the Python benchmark build the same particles.

    python tools/make_dam_config.py >> config/sph_scene.xml      (scene blocks on stdout, tank files written)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "..")))

from fluidsolver_amd import scenes  # noqa: E402

SCENES = {20: ("dam128k", (64, 32, 64)), 21: ("dam1m", (128, 64, 128)), 22: ("dam16m", (256, 256, 256))}


def fmt(v):
    return ", ".join(repr(float(x)) for x in v)


def main():
    for sid, (name, (nx, ny, nz)) in SCENES.items():
        g = scenes.dam_break_geometry(nx, ny, nz)
        P = g["params"]
        tank_file = f"{name} tank.xml"
        print(f'''
<MultiScene id="{sid}">
	<Name> Dam break {nx}x{ny}x{nz} (WCSPH) </Name>
	<MaxPNum> {nx * ny * nz * 2} </MaxPNum>
	<TypeNum> 3 </TypeNum>
	<DT> {P["dt"]!r} </DT>
	<PSpacing> {P["spacing"]!r} </PSpacing>
	<SmoothRatio> {P["smoothratio"]!r} </SmoothRatio>
	<Viscosity> {P["viscosity"]!r} </Viscosity>
	<RestDensity> 1000 </RestDensity>
	<PressureK> {P["pressureK"]!r} </PressureK>
	<DensityArray> 1000,1000,1000 </DensityArray>
	<ViscosityArray> 1000,1000,1000 </ViscosityArray>
	<GridMin> {fmt(P["vmin"])} </GridMin>
	<GridMax> {fmt(P["vmax"])} </GridMax>
	<WallType> 2 </WallType>
	<WallRestDensity> 1000 </WallRestDensity>
	<BoundaryMode> {P["boundary_mode"]} </BoundaryMode>
	<BoundaryFile> {tank_file} </BoundaryFile>
	<Vol>
		<VolMin> {fmt(g["fluid_lo"])} </VolMin>
		<VolMax> {fmt(g["fluid_hi"])} </VolMax>
		<VolFrac> 1, 0, 0 </VolFrac>
		<Group> 0 </Group>
	</Vol>
</MultiScene>''')
        with open(os.path.join(HERE, "..", "config", tank_file), "w") as f:
            f.write(f'''<!-- Tank of the {name} scene: floor and four walls, one layer of particles. -->
<Boundary>
    <OpenBox>
        <xmin> {fmt(g["tank_lo"])} </xmin>
        <xmax> {fmt(g["tank_hi"])} </xmax>
        <spacing> {P["spacing"]!r} </spacing>
        <id> 1 </id>
    </OpenBox>
</Boundary>
''')


if __name__ == "__main__":
    main()
