const char afx_build_id_str[] = "4ba43b4b5f6d";
