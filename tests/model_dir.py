"""Builds a HiPIMS model directory for the Newcastle-centre example (config C1) in a temp folder: the DEM is the data
file of the reference's own example (tests/golden/NewcastleCentreDEM_2m.img, an RLE-compressed float32 HFA raster);
the XML and the two CSV series are written here from the example's parameters (342x195 @ 2 m, Godunov, Courant 0.5,
friction on, 70 mm/h rain for an hour then nothing, 12 mm/h drainage, four closed edges, outputs every 600 s)."""
import os
import shutil

from conftest import GOLDEN

XML = """<?xml version="1.0"?>
<configuration>
  <metadata><name>Newcastle centre 2 m (config C1)</name></metadata>
  <execution><executor name="HIP"><parameter name="deviceFilter" value="GPU" /></executor></execution>
  <simulation>
    <parameter name="duration" value="{duration}" />
    <parameter name="outputFrequency" value="{frequency}" />
    <parameter name="floatingPointPrecision" value="double" />
    <domainSet>
      <domain type="cartesian" deviceNumber="1">
        <data sourceDir="topography/" targetDir="output/">
          <dataSource type="constant" value="velocityX" source="0.0" />
          <dataSource type="constant" value="velocityY" source="0.0" />
          <dataSource type="constant" value="depth" source="0.0" />
          <dataSource type="constant" value="manningCoefficient" source="0.030" />
          <dataSource type="raster" value="structure,dem" source="NewcastleCentreDEM_2m.img" />
          <dataTarget type="raster" value="depth" format="HFA" target="depth_%t.img" />
          <dataTarget type="raster" value="velocityX" format="HFA" target="velX_%t.img" />
          <dataTarget type="raster" value="velocityY" format="HFA" target="velY_%t.img" />
          <dataTarget type="raster" value="fsl" format="HFA" target="fsl_%t.img" />
          <dataTarget type="raster" value="maxdepth" format="HFA" target="maxdepth_%t.img" />
        </data>
        <scheme name="{scheme}">
          <parameter name="courantNumber" value="0.50" />
          <parameter name="frictionEffects" value="yes" />
          <parameter name="groupSize" value="32x8" />
        </scheme>
        <boundaryConditions sourceDir="./">
          <domainEdge edge="north" treatment="closed" />
          <domainEdge edge="south" treatment="closed" />
          <domainEdge edge="east" treatment="closed" />
          <domainEdge edge="west" treatment="closed" />
          <timeseries type="atmospheric" name="Drainage" value="loss-rate" source="boundaries/drainage.csv" />
          <timeseries type="atmospheric" name="Rainfall" value="rain-intensity" source="boundaries/rainfall.csv" />
        </boundaryConditions>
      </domain>
    </domainSet>
  </simulation>
</configuration>
"""


def make_newcastle(tmpdir, duration=7200, frequency=600, scheme="Godunov"):
    root = str(tmpdir)
    os.makedirs(os.path.join(root, "topography"), exist_ok=True)
    os.makedirs(os.path.join(root, "boundaries"), exist_ok=True)
    shutil.copy(os.path.join(GOLDEN, "NewcastleCentreDEM_2m.img"), os.path.join(root, "topography"))
    with open(os.path.join(root, "boundaries", "rainfall.csv"), "w") as f:
        f.write("Time (s),Rainfall intensity (mm/hr)\n0,70\n3600,70\n7200,0\n10800,0\n14400,0\n18000,0\n")
    with open(os.path.join(root, "boundaries", "drainage.csv"), "w") as f:
        f.write("Time (s),Drainage losses (mm/hr)\n0,12\n100000000,12\n")
    path = os.path.join(root, "newcastle-centre.xml")
    with open(path, "w") as f:
        f.write(XML.format(duration=duration, frequency=frequency, scheme=scheme))
    return path
