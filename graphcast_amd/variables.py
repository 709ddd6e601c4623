"""Variable / level constants, same names and values as the reference's
``weathernext/utils/variables.py:17-82`` (data of the task definition)."""

PRESSURE_LEVELS_ERA5_37 = (
    1, 2, 3, 5, 7, 10, 20, 30, 50, 70, 100, 125, 150, 175, 200, 225, 250, 300,
    350, 400, 450, 500, 550, 600, 650, 700, 750, 775, 800, 825, 850, 875, 900,
    925, 950, 975, 1000)
PRESSURE_LEVELS_HRES_25 = (
    1, 2, 3, 5, 7, 10, 20, 30, 50, 70, 100, 150, 200, 250, 300, 400, 500, 600,
    700, 800, 850, 900, 925, 950, 1000)
PRESSURE_LEVELS_WEATHERBENCH_13 = (
    50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000)
PRESSURE_LEVELS = {13: PRESSURE_LEVELS_WEATHERBENCH_13, 25: PRESSURE_LEVELS_HRES_25,
                   37: PRESSURE_LEVELS_ERA5_37}

ALL_ATMOSPHERIC_VARS = (
    "potential_vorticity", "specific_rain_water_content", "specific_snow_water_content",
    "geopotential", "temperature", "u_component_of_wind", "v_component_of_wind",
    "specific_humidity", "vertical_velocity", "vorticity", "divergence", "relative_humidity",
    "ozone_mass_mixing_ratio", "specific_cloud_liquid_water_content",
    "specific_cloud_ice_water_content", "fraction_of_cloud_cover")
ALL_SURFACE_VARS = (
    "2m_temperature", "mean_sea_level_pressure", "10m_v_component_of_wind",
    "10m_u_component_of_wind", "total_precipitation_12hr", "total_precipitation_6hr",
    "sea_surface_temperature")
EXTERNAL_FORCING_VARS = ("toa_incident_solar_radiation",)
TIME_FORCING_VARS = ("year_progress_sin", "year_progress_cos", "day_progress_sin",
                     "day_progress_cos")
STATIC_VARS = ("geopotential_at_surface", "land_sea_mask")
